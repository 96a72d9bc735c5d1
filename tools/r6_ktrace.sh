#!/bin/bash
# round 6: kernel trace of the default bench command (headline run only) -> kernel statistics (launch counts per kernel) + concurrency summary. Usage: tools/r6_ktrace.sh <tag> [config] [steps]
set -u
TAG=$1; CFG=${2:-ont_hg38}; STEPS=${3:-25}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace -d $O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --cpu-sample 0 --verify 0 --extra-configs "" --no-host-input --steps $STEPS > $O/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name '*.db' | head -1)
python tools/trace_db.py $DB --csv gpurun_out/${TAG}_kernel_stats.csv --timed $STEPS > gpurun_out/${TAG}_concurrency.txt 2>&1
cp $O/bench_prof.json gpurun_out/${TAG}_bench_line_under_rocprof.json
rm -rf $O/prof
head -12 gpurun_out/${TAG}_concurrency.txt
python - <<P
import json
d = json.loads([l for l in open('$O/bench_prof.json') if l.startswith('{')][-1])
print('$CFG', 'value', round(d['value'], 3), 'ms/step', round(d['ms_per_step'], 2), 'syncs', d['host_syncs_per_step'], 'cores', d['host_cores_busy_timed_pass'])
P
