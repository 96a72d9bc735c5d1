#!/bin/bash
# On the GPU box (round 5, first call): GPU tests, then the chain kernels one wavefront per read (VMX_CHAIN_ROWS=0) against four reads per wavefront (default)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5a_gputest.log 2>&1; tail -5 gpurun_out/r5a_gputest.log
bash tools/envsweep.sh "VMX_CHAIN_ROWS=0" "" "VMX_CHAIN_ROWS=0" "" 2>&1 | tee gpurun_out/r5a_ab_rows.txt
VMX_CHAIN_ROWS=0 bash tools/prof1.sh r5a_old --extra-configs "" > gpurun_out/r5a_prof1_old.txt 2>&1
bash tools/prof1.sh r5a_rows --extra-configs "" > gpurun_out/r5a_prof1_rows.txt 2>&1
grep -E "chain|span" gpurun_out/r5a_prof1_old.txt gpurun_out/r5a_prof1_rows.txt
