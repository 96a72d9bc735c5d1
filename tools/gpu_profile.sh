#!/bin/bash
# On the GPU box: the default bench line, its rocprofv3 kernel trace (3 batches in flight and 1), and the three PMC passes of the same
# command (separate passes, --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes). Usage: tools/gpu_profile.sh <tag> [pmc]
# Output under gpurun_out/<tag>/ ; tools/pmc_traffic.py and tools/trace_db.py turn it into the profiles/ summaries.
set -u
TAG=${1:-r03}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/prof3 -o p3 -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --verify 0 --extra-configs "" > $OUT/bench_prof3.json 2>/dev/null
rocprofv3 --kernel-trace -d $OUT/prof1 -o p1 -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --verify 0 --extra-configs "" --streams 1 --steps 8 > $OUT/bench_prof1.json 2>/dev/null
if [ "${2:-}" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    N=$(echo $C | cut -d' ' -f1)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --verify 0 --extra-configs "" > $OUT/bench_pmc_$N.json 2>/dev/null
    # keep only the counter table (the trace CSVs are large)
    find $OUT/pmc_$N -name '*kernel_trace.csv' -delete
  done
fi
ls -la $OUT
