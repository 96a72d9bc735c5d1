#!/bin/bash
# On the GPU box: the evidence set of the final code of round 6 (GPU tests, default bench line, profile sets per workload, bigverify, launch sequence, long driver run)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_z_gputest_42_tests.log 2>&1; tail -2 gpurun_out/r06_z_gputest_41_tests.log
timeout 1500 python bench.py > gpurun_out/r06_z_bench_line_default.json 2> gpurun_out/r6z/bench.err; tail -c 300 gpurun_out/r06_z_bench_line_default.json; echo
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_z_bench_line_driver_command_steps20_warmup5.json 2>> gpurun_out/r6z/bench.err
for c in ont_hg38 hifi_hg38 vacsim_r ont_100mb; do bash tools/gpu_profile_r6.sh r06_z_$c $c > gpurun_out/r6z/prof_$c.log 2>&1; tail -1 gpurun_out/r6z/prof_$c.log; done
bash tools/r6_seq.sh r06_z > /dev/null 2>&1
BIGVERIFY_SEED_OFFSET=8000 timeout 1500 python tools/bigverify.py > gpurun_out/r06_z_bigverify_16300_reads_seed8000.log 2>&1; tail -3 gpurun_out/r06_z_bigverify_16300_reads_seed8000.log
bash tools/r5_long.sh 10 16 _r6 > gpurun_out/r6z/long.log 2>&1; tail -5 gpurun_out/r6z/long.log
