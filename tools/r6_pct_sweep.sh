#!/bin/bash
# On the GPU box: the band-width rule of the gap fill (VMX_AD_PCT / VMX_AD_PCT_MIN) swept per config.  bash tools/r6_pct_sweep.sh "vacsim_r ont_hg38" "40:40 60:50 ..."
CFGS=$1; PCTS=$2
mkdir -p gpurun_out; : > gpurun_out/pct_sweep.txt
for cfg in $CFGS; do
 for pp in $PCTS; do
    P=${pp%%:*}; M=${pp#*:}
    VMX_AD_PCT=$P VMX_AD_PCT_MIN=$M timeout 500 python bench.py --config $cfg --extra-configs "" --cpu-sample 0 --no-host-input --verify 16 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['kernels']
print('$cfg', 'pct $P min $M', round(d['value'], 3), 'Gbp/s', round(d['ms_per_step'], 2), 'ms/step', 'fill', round(k['k_gapfill_fill_ns']['ms_per_step'], 2), 'trace', round(d['gapfill_trace_ms_per_step'], 2), d['oracle_crosscheck'], 'redo', round(d['dp_redo_per_step']), 'redo GB', round(d['dp_redo_tb_bytes_per_step']/1e9,2), 'cells/read', round(d['per_read']['dp_cells']))" >> gpurun_out/pct_sweep.txt
 done
done
cat gpurun_out/pct_sweep.txt
