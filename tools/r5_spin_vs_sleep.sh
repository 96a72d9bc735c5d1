#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "cpus: $(nproc), cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
run() { env $2 python bench.py --config $1 --steps 24 --warmup 5 --cpu-sample 0 --verify 0 --no-host-input --extra-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 $2:', round(d['ms_per_step'],3), 'ms', round(d['value'],3), 'Gbp/s, seed', round(d['stage_ms_per_step'][0],1), 'host active', round(d['host_active_ms_per_batch'],2))"; }
for c in ont_hg38 vacsim_r; do run $c VMX_BLOCKING_SYNC=0; run $c VMX_BLOCKING_SYNC=1; run $c VMX_BLOCKING_SYNC=0; run $c VMX_BLOCKING_SYNC=1; done
