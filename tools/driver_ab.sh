#!/bin/bash
# On the GPU box: the driver on the 393 k-read FASTQ of tools/driver_bench.py under different settings of one environment knob, alternating.
#   bash tools/driver_ab.sh VMX_DRIVER_WINDOWS 3 5 [extra driver args]
KNOB=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out
T=/tmp/vmx_driver_bench
if [ ! -f $T/reads.fq ]; then python tools/driver_bench.py --reads 393216 --out gpurun_out/driver_bench.json > gpurun_out/driver_bench.log 2>&1; fi
: > gpurun_out/driver_ab.txt
for rep in 1 2; do
  for v in $A $B; do
    sleep 12
    env $KNOB=$v VMX_DRIVER_TIMING=1 python -m vacmap_amd.driver -ref $T/ref.fa -read $T/reads.fq -mode H -o $T/out.sam -t 16 --nowriteindex --force "$@" 2>&1 | grep "vacmapx timing" | sed "s/^/$KNOB=$v /" >> gpurun_out/driver_ab.txt
  done
done
cat gpurun_out/driver_ab.txt
