import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, time
import oracle_lib as O, kernel_cases as KC
from vacmap_amd.lib import Context
ctx = Context(0)
case = sys.argv[1]
rng = np.random.default_rng(1)
def mk(m):
    a = KC.rand_seq(rng, m); return a, KC.mutate(rng, a, 0.1)
if case == 'two':
    ps = [mk(300), mk(500)]
elif case == 'empty':
    ps = [mk(300), ('ACGT', ''), mk(200)]
elif case == 'emptyq':
    ps = [mk(300), ('', 'ACGT'), mk(200)]
elif case == 'many':
    ps = [mk(int(rng.integers(10, 800))) for _ in range(int(sys.argv[2]))]
qs = [p[0] for p in ps]; ts = [p[1] for p in ps]
t = time.time(); g = ctx.edit_distance_batch(qs, ts); print(case, g.tolist()[:8], [O.edit_distance(a,b) for a,b in ps][:8], time.time()-t)
