#!/bin/bash
# On the GPU box: k_local_seed_band alone (one batch in flight) for a few tile-size / register-budget sets, ONT and HiFi configs.
# Usage: tools/lb_sweep.sh <tag> "name:flags" ...
set -u
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for V in "$@"; do
  N=${V%%:*}; F=${V#*:}
  VMX_EXTRA_FLAGS="$F" timeout 300 python -m vacmap_amd.build --force > /dev/null 2>&1 || { echo "$N build failed"; continue; }
  A=$(tools/prof1.sh ${TAG}_$N --extra-configs "" | grep "k_local_seed_band" | tail -1)
  B=$(tools/prof1.sh ${TAG}_${N}_hifi --config hifi_hg38 --extra-configs "" | grep "k_local_seed_band" | tail -1)
  echo "$N | ONT $A | HiFi $B" | tee -a $OUT/sweep.txt
done
python -m vacmap_amd.build --force > /dev/null 2>&1
