#!/bin/bash
# On the GPU box: LDS counters per kernel of one bench workload (separate --pmc passes, --kernel-trace only). Usage: tools/r6_lds_pmc.sh <config> [steps]
set -u
CFG=$1; STEPS=${2:-6}
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/lds_$CFG; mkdir -p $O
B="--config $CFG --cpu-sample 0 --verify 0 --extra-configs \"\" --no-host-input --steps $STEPS --warmup 2"
cd /tmp && export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  eval timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py $B > $O/bench_$N.json 2> $O/err_$N.txt
  find $O/$N -name '*kernel_trace.csv' -delete
done
cd $GRAFT_REPO_ROOT
python - <<P > gpurun_out/r06_v_lds_counters_${CFG}.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob('$O/*/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
names = sorted({c for k in tot for c in tot[k]})
print('LDS counters per kernel, workload $CFG, summed over every launch of $STEPS + 2 batches (rocprofv3 --pmc, one pass per counter pair)')
print('%-28s' % 'kernel' + ''.join('%24s' % n for n in names) + '   conflict / idx_active')
for k in sorted(tot, key=lambda k: -tot[k].get('SQ_LDS_IDX_ACTIVE', 0))[:14]:
    t = tot[k]
    print('%-28s' % k[:28] + ''.join('%24.4g' % t.get(n, 0) for n in names) + '   %.3f' % (t.get('SQ_LDS_BANK_CONFLICT', 0) / max(t.get('SQ_LDS_IDX_ACTIVE', 0), 1)))
P
cat gpurun_out/r06_v_lds_counters_${CFG}.txt | cut -c1-260
rm -rf $O/SQ_*
