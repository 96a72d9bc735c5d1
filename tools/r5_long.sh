#!/bin/bash
# On the GPU box: the command-line driver on a LONG input (VERDICT r4 missing 4): 163840 generated ONT reads written ten times under different names
# = 1.6 M reads, ~49 GB of FASTQ in /dev/shm, SAM out next to it; per-100k-read rates from the driver's own progress lines.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "memory.max: $(cat /sys/fs/cgroup/memory.max 2>/dev/null)"
REP=${1:-10}; T=${2:-16}; TAG=${3:-}; SAMDIR=${SAMDIR:-/dev/shm/vmx_long}
timeout 1500 python tools/driver_bench.py --reads 163840 --replicate $REP --t $T --tmp /dev/shm/vmx_long --sam-dir $SAMDIR "--driver-args=${DARGS:-}" --out gpurun_out/r5_driver_long_${REP}x_t$T$TAG.json 2> gpurun_out/r5_driver_long_${REP}x_t$T$TAG.err | tail -c 3000
tail -4 gpurun_out/r5_driver_long_${REP}x_t$T$TAG.err
rm -rf /dev/shm/vmx_long $SAMDIR
