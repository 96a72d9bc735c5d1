#!/bin/bash
# On the GPU box: one PMC pass of the bench (one batch in flight, a few steps) with the given counters, summed per kernel.
# Usage: tools/pmc_kernel.sh <tag> "<COUNTER ...>" [kernel-name-substring]    -> gpurun_out/<tag>/pmc_<first counter>.txt
set -u
TAG=$1; CTRS=$2; KSUB=${3:-k_local_seed}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=$(echo $CTRS | cut -d' ' -f1)
timeout 500 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --verify 0 --streams 1 --steps 4 > /dev/null 2>&1
python - "$KSUB" /tmp/pmc_${TAG}_$N > $OUT/pmc_$N.txt <<'PY'
import csv, glob, sys, collections
sub, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    if sub not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); calls[(k, r['Counter_Name'])] += 1
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()): print('   %-24s %.4g  (%d dispatches)' % (c, v, calls[(k, c)]))
PY
cat $OUT/pmc_$N.txt
