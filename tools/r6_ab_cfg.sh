#!/bin/bash
# On the GPU box: bench of several configs with two builds of the library in alternation.  bash tools/r6_ab_cfg.sh _ab/libvacmapx_x.so "hifi_hg38 vacsim_r ont_hg38" [rounds]
OLD=$1; CFGS=$2; N=${3:-1}
mkdir -p gpurun_out; : > gpurun_out/ab_cfg.txt
for i in $(seq 1 $N); do
 for cfg in $CFGS; do
  for which in old new; do
    if [ $which = old ]; then export VACMAPX_LIB=$PWD/$OLD; else unset VACMAPX_LIB; fi
    timeout 500 python bench.py --config $cfg --extra-configs "" --cpu-sample 0 --no-host-input --verify 16 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['kernels']
print('$cfg', '$which', round(d['value'], 3), 'Gbp/s', round(d['ms_per_step'], 2), 'ms/step', 'fill', round(k['k_gapfill_fill_ns']['ms_per_step'], 2), 'trace', round(d['gapfill_trace_ms_per_step'], 2), 'lseed', round(k['k_local_seed']['ms_per_step'], 2), 'ctx', d['config']['schedule'][-25:], 'hbm', round(d['hbm_used_gb']), d['oracle_crosscheck'], 'redo', round(d['dp_redo_per_step']), 'cells/read', round(d['per_read']['dp_cells']))" >> gpurun_out/ab_cfg.txt
  done
 done
done
cat gpurun_out/ab_cfg.txt
