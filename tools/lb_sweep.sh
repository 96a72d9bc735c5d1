#!/bin/bash
# On the GPU box: k_local_seed_band alone (one batch in flight) for a few tile-size / register-budget sets, ONT and HiFi configs.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
for V in "q1024:-DVMX_LB_QC=1024 -DVMX_LB_HCAP=1024 -DVMX_LB_SORTK=1024 -DVMX_LB_NBLOG=10" "q768:-DVMX_LB_QC=768 -DVMX_LB_HCAP=768 -DVMX_LB_SORTK=1024 -DVMX_LB_NBLOG=10" "q512h768:-DVMX_LB_QC=512 -DVMX_LB_HCAP=768 -DVMX_LB_SORTK=1024 -DVMX_LB_NBLOG=9"; do
  N=${V%%:*}; F=${V#*:}
  VMX_EXTRA_FLAGS="$F" timeout 300 python -m vacmap_amd.build --force > /dev/null 2>&1 || { echo "$N build failed"; continue; }
  A=$(tools/prof1.sh ${1:-sweep}_$N --extra-configs "" | grep "k_local_seed_band" | tail -1)
  B=$(tools/prof1.sh ${1:-sweep}_${N}_hifi --config hifi_hg38 --extra-configs "" | grep "k_local_seed_band" | tail -1)
  echo "$N | ONT $A | HiFi $B" | tee -a $OUT/sweep.txt
done
