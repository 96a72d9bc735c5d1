#!/bin/bash
# On the GPU box: the bases one pass over the path takes (VMX_MAX_BATCH_BASES, default 160 Mi) - the longest-read batch of the ONT workload holds 174.7 M and ran as 167.8 M + a remainder
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 600 python bench.py --extra-configs "" --cpu-sample 0 --no-host-input --verify $V 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value'],3), round(d['ms_per_step'],2), 'ctx', d['contexts_in_flight'], 'small', d['small_contexts_added'], 'HBM', round(d['hbm_used_gb'],1), 'waits', d['host_syncs_per_step'], d['oracle_crosscheck'])"; }
V=200; run VMX_MAX_BATCH_BASES=200000000
V=16; run VMX_X=0
run VMX_MAX_BATCH_BASES=200000000
run VMX_X=0
run VMX_MAX_BATCH_BASES=200000000
