#!/bin/bash
# How much would MORE full contexts give on an ONT-shape workload if they fitted? Batches of 2048 reads (pools of a full context half as large),
# so that 5 .. 8 full contexts fit; same batches for every line.
cd $GRAFT_REPO_ROOT
for spec in "5 1" "6 1" "7 1" "8 1" "6 0" "5 1"; do
  set -- $spec
  VMX_FULL_CTX=$1 VMX_NO_FULL_GROWTH=1 VMX_SMALL_CTX=$2 VMX_MAX_CTX=$(( $1 + $2 )) python bench.py --steps 80 --warmup 5 --reads-per-step 2048 --extra-configs "" --cpu-sample 0 --verify 0 --no-host-input 2>/tmp/probe.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('full $1 small $2:', round(d['ms_per_step'],3), 'ms', round(d['value'],3), 'Gbp/s', d.get('small_contexts_added'), 'small added,', round(d['hbm_used_gb'],1), 'GB, given up', d.get('contexts_given_up_for_memory'))"
done
