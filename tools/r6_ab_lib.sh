#!/bin/bash
# On the GPU box: the headline bench with two builds of the library in alternation (VACMAPX_LIB), same box, same run.  bash tools/r6_ab_lib.sh _ab/libvacmapx_x.so [rounds] [extra bench args]
OLD=$1; N=${2:-2}; shift; shift
mkdir -p gpurun_out; : > gpurun_out/ab_lib.txt
for i in $(seq 1 $N); do
  for which in old new; do
    if [ $which = old ]; then export VACMAPX_LIB=$PWD/$OLD; else unset VACMAPX_LIB; fi
    python bench.py --extra-configs "" --cpu-sample 0 --no-host-input --verify 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', round(d['value'], 3), 'Gbp/s', round(d['ms_per_step'], 2), 'ms/step', 'syncs', d['host_syncs_per_step'], 'cores', d['host_cores_busy_timed_pass'], [round(x, 1) for x in d['stage_ms_per_step']])" >> gpurun_out/ab_lib.txt
  done
done
cat gpurun_out/ab_lib.txt
