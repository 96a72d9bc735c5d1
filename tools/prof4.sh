#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of the default bench (4 batches in flight), summarised: per-kernel stats csv + concurrency / wall share.
# Usage: tools/prof4.sh <tag> [extra bench args]     -> gpurun_out/<tag>/{kernel_stats.csv,concurrency.txt,bench_prof4.json}
set -u
TAG=${1:-p4}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof4_$TAG -o p4 -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --verify 0 --extra-configs "" "$@" > $OUT/bench_prof4.json 2>/dev/null
DB=$(find /tmp/prof4_$TAG -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/trace_db.py $DB --csv $OUT/kernel_stats.csv > $OUT/concurrency.txt 2>&1
cat $OUT/concurrency.txt
