#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of the bench with ONE batch in flight, summarised per kernel (ms per batch alone).
# Usage: tools/prof1.sh <tag> [extra bench args]     -> gpurun_out/<tag>/trace_summary_1stream.txt
set -u
TAG=${1:-p}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof1_$TAG -o p1 -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --verify 0 --streams 1 --steps 8 "$@" > $OUT/bench_prof1.json 2>/dev/null
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof1_$TAG > $OUT/trace_summary_1stream.txt 2>&1
tail -28 $OUT/trace_summary_1stream.txt
