#!/bin/bash
# On the GPU box: the default bench (no side configs, no CPU leg) with two builds of the library in alternation (VACMAPX_LIB), same box.
#   bash tools/ab_lib2.sh _ab/libvacmapx_old.so [rounds] [extra bench args]
OLD=$1; N=${2:-3}; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; : > gpurun_out/ab_lib.txt
for i in $(seq 1 $N); do
  for which in old new; do
    if [ $which = old ]; then export VACMAPX_LIB=$PWD/$OLD; else unset VACMAPX_LIB; fi
    timeout 400 python bench.py --extra-configs "" --cpu-sample 0 --verify 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', round(d['value'], 3), 'Gbp/s', round(d['ms_per_step'], 2), 'ms/step')" >> gpurun_out/ab_lib.txt
  done
done
cat gpurun_out/ab_lib.txt
