#!/bin/bash
# On the GPU box: one batch in flight under rocprofv3 --kernel-trace; durations of the launches of one kernel by their position inside a batch
# (k_ext_phase runs nine times per batch: which phase is the long one?).   bash tools/phase_times.sh [kernel] [launches per batch]
K=${1:-k_ext_phase}; PER=${2:-9}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pt -o pt -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - "$K" "$PER" <<'PY'
import sqlite3, sys, glob, collections
k, per = sys.argv[1], int(sys.argv[2])
db = glob.glob('gpurun_out/pt/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute('select name, start, end from kernels order by start').fetchall()
rows = [(n.split('(')[0], s, e) for n, s, e in rows]
sk = [r[1] for r in rows if r[0] == 'k_sketch']
t0 = sk[4] if len(sk) > 4 else rows[0][1]
calls = [r for r in rows if r[0] == k and r[1] >= t0]
agg = collections.defaultdict(list)
for i, r in enumerate(calls):
    agg[i % per].append((r[2] - r[1]) / 1e6)
for p in sorted(agg):
    v = agg[p]; print('%s call %d of a batch: mean %.3f ms  max %.3f ms  (%d batches)' % (k, p, sum(v) / len(v), max(v), len(v)))
PY
rm -rf gpurun_out/pt
