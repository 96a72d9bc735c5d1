#!/bin/bash
# On the GPU box: GPU tests of the chain kernels, A/B of VMX_CHAIN_ROWS, one-stream kernel profile
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5b_gputest.log 2>&1; tail -3 gpurun_out/r5b_gputest.log
bash tools/envsweep.sh "VMX_CHAIN_ROWS=0" "" "VMX_CHAIN_ROWS=0" "" 2>&1 | tee gpurun_out/r5b_ab_rows.txt
bash tools/prof1.sh r5b_rows --extra-configs "" > gpurun_out/r5b_prof1_rows.txt 2>&1
head -3 gpurun_out/r5b_rows/trace_summary_1stream.txt; grep -E "chain|span" gpurun_out/r5b_prof1_rows.txt | tail -5
bash tools/prof1.sh r5b_rows_hifi --config hifi_hg38 --extra-configs "" > gpurun_out/r5b_prof1_rows_hifi.txt 2>&1
tail -14 gpurun_out/r5b_prof1_rows_hifi.txt
