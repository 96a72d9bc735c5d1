#!/bin/bash
# On the GPU box: the driver with the host-pipeline changes of this round switched off / on as a set (pinned read blobs, heap pages kept, writev writer).
mkdir -p gpurun_out
T=/tmp/vmx_driver_bench
if [ ! -f $T/reads.fq ]; then python tools/driver_bench.py --reads 393216 --out gpurun_out/driver_bench.json > gpurun_out/driver_bench.log 2>&1; fi
: > gpurun_out/driver_ab2.txt
for rep in 1 2; do
  for v in 0 1; do
    sleep 12
    env VMX_DRIVER_PINNED=$v VMX_DRIVER_MALLOPT=$v VMX_DRIVER_WRITEV=$v VMX_DRIVER_TIMING=1 python -m vacmap_amd.driver -ref $T/ref.fa -read $T/reads.fq -mode H -o $T/out$v.sam -t 16 --nowriteindex --force 2>&1 | grep "vacmapx timing" | sed "s/^/host pipeline changes=$v /" >> gpurun_out/driver_ab2.txt
  done
done
cmp <(grep -v "^@PG" $T/out0.sam) <(grep -v "^@PG" $T/out1.sam) && echo "SAM files identical (all lines but @PG, which holds the output path)" >> gpurun_out/driver_ab2.txt
cat gpurun_out/driver_ab2.txt
python -c "
import json; d = json.load(open('gpurun_out/driver_bench.json')); print('driver_bench (new defaults): loop %.2f s, whole %.2f s, resident %.2f s, loop/resident %.2f' % (d['driver_loop_s'], d['driver_wall_s'], d['resident_pipeline_s'], d['driver_loop_over_resident']))"
