#!/bin/bash
# On the GPU box: the scheduler's bases limit per job (VMX_BATCH_MAX_BASES) swept on bench configs.  bash tools/r6_max_bases.sh "cfg ..." "limit ..." [steps]
CFGS=$1; LIMS=$2; STEPS=${3:-25}
mkdir -p gpurun_out; : > gpurun_out/max_bases.txt
for cfg in $CFGS; do
 for L in $LIMS; do
    VMX_BATCH_MAX_BASES=$L timeout 700 python bench.py --config $cfg --steps $STEPS --extra-configs "" --cpu-sample 0 --no-host-input --verify 16 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', 'limit', '$L', 'steps', d['steps'], round(d['value'], 3), 'Gbp/s', round(d['ms_per_step'], 2), 'ms/step', 'jobs/step', round(d['jobs_per_step'], 2), 'largest job', round(d['largest_job_bases'] / 1e6, 1), 'Mb', 'contexts', d['contexts_in_flight'],
      '(+%d full, %d small)' % (d['contexts_added_for_memory'], d['small_contexts_added']), 'HBM', round(d['hbm_used_gb'], 1), 'GB', 'waits/step', d['host_syncs_per_step'], 'cores', d['host_cores_busy_timed_pass'], d['oracle_crosscheck'])
" | tee -a gpurun_out/max_bases.txt
 done
done
