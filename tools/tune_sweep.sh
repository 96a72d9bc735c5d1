#!/bin/bash
# On the GPU box: the default bench command under a list of launch-width knobs (one line per setting in gpurun_out/<tag>/sweep.txt).
TAG=${1:-sweep}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : " >> $OUT/sweep.txt; env "$@" python bench.py --cpu-sample 0 --verify 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), round(d['ms_per_step'],2), round(d['hbm_used_gb']))" >> $OUT/sweep.txt; }
run X=0
run VMX_FILL_WAVES=8
run VMX_FILL_WAVES=12
run VMX_FILL_WAVES=24
run VMX_LSEED_WGS=3
run VMX_LSEED_WGS=1
run VMX_GC_LDS_MAX=0 VMX_LC_LDS_MAX=0
run VMX_GC_LDS_MAX=1024 VMX_LC_LDS_MAX=1024
run VMX_EXT_SPREAD=4
run VMX_EXT_SPREAD=16
run VMX_TRACE_SPREAD=4
run GPU_MAX_HW_QUEUES=24
run X=1
cat $OUT/sweep.txt
