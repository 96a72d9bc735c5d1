#!/bin/bash
# On the GPU box: the default bench with two builds of the library in alternation (VACMAPX_LIB), same box, same run.
#   bash tools/ab_lib.sh _ab/libvacmapx_old.so [rounds]
OLD=$1; N=${2:-3}
mkdir -p gpurun_out; : > gpurun_out/ab_lib.txt
for i in $(seq 1 $N); do
  for which in old new; do
    if [ $which = old ]; then export VACMAPX_LIB=$PWD/$OLD; else unset VACMAPX_LIB; fi
    python bench.py --steps 16 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', round(d['value'], 3), 'Gbp/s', round(d['ms_per_step'], 2), 'ms/step', [round(x, 1) for x in d['stage_ms_per_step']])" >> gpurun_out/ab_lib.txt
  done
done
cat gpurun_out/ab_lib.txt
