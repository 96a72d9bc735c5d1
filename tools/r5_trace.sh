#!/bin/bash
# On the GPU box: rocprofv3 kernel traces of one bench workload with the default number of batches in flight and with one -> kernel statistics + concurrency summaries
# Usage: tools/r5_trace.sh <tag> <config> [steps]
set -u
TAG=$1; CFG=$2; STEPS=${3:-15}
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/prof3 -o p3 -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --cpu-sample 0 --verify 0 --extra-configs "" --no-host-input --steps $STEPS > $O/bench_prof3.json 2>/dev/null
timeout 900 rocprofv3 --kernel-trace -d $O/prof1 -o p1 -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --cpu-sample 0 --verify 0 --extra-configs "" --no-host-input --streams 1 --steps 8 > $O/bench_prof1.json 2>/dev/null
cd $GRAFT_REPO_ROOT
DB3=$(find $O/prof3 -name '*.db' | head -1); DB1=$(find $O/prof1 -name '*.db' | head -1)
python tools/trace_db.py $DB3 --csv gpurun_out/${TAG}_kernel_stats_5streams.csv --skip 5 > gpurun_out/${TAG}_concurrency_5streams.txt 2>&1
python tools/trace_db.py $DB1 --csv gpurun_out/${TAG}_kernel_stats_1stream.csv --skip 1 > gpurun_out/${TAG}_concurrency_1stream.txt 2>&1
rm -rf $O/prof3 $O/prof1
head -34 gpurun_out/${TAG}_concurrency_5streams.txt
