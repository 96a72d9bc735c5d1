#!/bin/bash
# On the GPU box: bench configs with several builds of the library (VACMAPX_LIB).  bash tools/r6_lib_variants.sh "cfg ..." lib1.so lib2.so ...   ("base" = the in-tree build)
CFGS=$1; shift
mkdir -p gpurun_out; : > gpurun_out/lib_variants.txt
for cfg in $CFGS; do
 for L in "$@"; do
    if [ $L = base ]; then unset VACMAPX_LIB; else export VACMAPX_LIB=$PWD/$L; fi
    timeout 500 python bench.py --config $cfg --extra-configs "" --cpu-sample 0 --no-host-input --verify 16 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['kernels']
print('$cfg', '$L', round(d['value'], 3), 'Gbp/s', round(d['ms_per_step'], 2), 'ms/step', 'lseed', round(k['k_local_seed']['ms_per_step'], 2), 'fill', round(k['k_gapfill_fill_ns']['ms_per_step'], 2), d['oracle_crosscheck'], 'general', d['local_general_reads'])" >> gpurun_out/lib_variants.txt
 done
done
cat gpurun_out/lib_variants.txt
