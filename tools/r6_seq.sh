#!/bin/bash
# On the GPU box: the time-ordered launches of ONE batch (one batch in flight), csv kernel trace.  tools/r6_seq.sh <tag> [config] [batch index]
set -u
TAG=$1; CFG=${2:-ont_hg38}; BI=${3:-3}
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/seq -o s -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --cpu-sample 0 --verify 0 --extra-configs "" --no-host-input --streams 1 --steps 5 > $O/bench_seq.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/trace_sequence.py $O/seq $BI > gpurun_out/${TAG}_sequence_one_batch.txt
rm -rf $O/seq
wc -l gpurun_out/${TAG}_sequence_one_batch.txt
