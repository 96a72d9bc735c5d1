cd $GRAFT_REPO_ROOT
run() { env "$@" VMX_DBG_POOLS=1 timeout 400 python bench.py --extra-configs "" --cpu-sample 0 --no-host-input --verify 0 2> gpurun_out/mb.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value'],3), round(d['ms_per_step'],2), 'ctx', d['contexts_in_flight'], 'HBM', round(d['hbm_used_gb'],1), 'waits', d['host_syncs_per_step'], 'jobs/step', d['jobs_per_step'])"; grep "process so far" gpurun_out/mb.err | tail -1; }
run VMX_BATCH_MAX_BASES=0
run VMX_BATCH_MAX_BASES=100000000 VMX_MAX_FULL_CTX=6 VMX_SMALL_CTX=0
run VMX_BATCH_MAX_BASES=100000000 VMX_MAX_FULL_CTX=7 VMX_SMALL_CTX=0
run VMX_BATCH_MAX_BASES=90000000 VMX_MAX_FULL_CTX=6 VMX_SMALL_CTX=0
run VMX_BATCH_MAX_BASES=0
run VMX_BATCH_MAX_BASES=100000000 VMX_MAX_FULL_CTX=6 VMX_SMALL_CTX=0
