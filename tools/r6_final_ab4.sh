#!/bin/bash
# On the GPU box: profile sets of the two hg38 workloads and a bigverify on the last commit of round 6 (traceback lines of 4 anti-diagonals)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6zz
for c in ont_hg38 hifi_hg38; do bash tools/gpu_profile_r6.sh r06_zzz_$c $c > gpurun_out/r6zz/prof_$c.log 2>&1; tail -1 gpurun_out/r6zz/prof_$c.log; done
BIGVERIFY_SEED_OFFSET=10000 timeout 1500 python tools/bigverify.py > gpurun_out/r06_zzz_bigverify_16300_reads_seed10000.log 2>&1; tail -3 gpurun_out/r06_zzz_bigverify_16300_reads_seed10000.log
