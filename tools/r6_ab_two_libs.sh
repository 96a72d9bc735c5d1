#!/bin/bash
# On the GPU box: default bench line with two builds of the library, alternating.  bash tools/r6_ab_two_libs.sh a.so b.so [rounds] [extra bench args]
A=$1; B=$2; R=${3:-2}; shift 3
cd $GRAFT_REPO_ROOT; : > gpurun_out/ab_two_libs.txt
for i in $(seq $R); do for L in $A $B; do
  VACMAPX_LIB=$PWD/$L timeout 600 python bench.py --extra-configs "" --cpu-sample 0 --no-host-input --verify 16 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],3), round(d['ms_per_step'],2), 'ctx', d['contexts_in_flight'], 'small', d['small_contexts_added'], 'HBM', round(d['hbm_used_gb'],1), 'waits', d['host_syncs_per_step'], d['oracle_crosscheck'])" | tee -a gpurun_out/ab_two_libs.txt
done; done
