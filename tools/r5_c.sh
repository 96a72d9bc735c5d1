#!/bin/bash
# On the GPU box: the default library against a variant (VACMAPX_LIB) — one-stream kernel profile of each (chain kernels per batch) + alternating bench runs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
V=$1
bash tools/prof1.sh r5c_def --extra-configs "" > /dev/null 2>&1
VACMAPX_LIB=$PWD/$V bash tools/prof1.sh r5c_var --extra-configs "" > /dev/null 2>&1
for t in r5c_def r5c_var; do echo $t; head -2 gpurun_out/$t/trace_summary_1stream.txt | cut -c1-200; grep -E "k_chain|mean span" gpurun_out/$t/trace_summary_1stream.txt; done
bash tools/ab_lib2.sh $V 2
VMX_DBG_CHAIN=1 timeout 300 python bench.py --extra-configs "" --cpu-sample 0 --verify 0 --streams 1 --steps 4 2>&1 | grep "chain rows" | tail -2
