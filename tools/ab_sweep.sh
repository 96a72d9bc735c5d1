#!/bin/bash
# On the GPU box: a build flag's values compared — per-kernel ms per batch alone (tools/prof1.sh) and the default bench step.
# Usage: tools/ab_sweep.sh <tag> <kernel-regex> "name:flags" ...
set -u
cd $GRAFT_REPO_ROOT
TAG=$1; KRE=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for V in "$@"; do
  N=${V%%:*}; F=${V#*:}
  VMX_EXTRA_FLAGS="$F" timeout 300 python -m vacmap_amd.build --force > /dev/null 2>&1 || { echo "$N build failed"; continue; }
  A=$(tools/prof1.sh ${TAG}_$N --extra-configs "" | grep -E "$KRE|mean span" | tr -s ' ' | tr '\n' ';')
  B=$(timeout 300 python bench.py --steps 24 --cpu-sample 0 --verify 16 --extra-configs "" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(round(d['ms_per_step'],2),d['oracle_crosscheck'])")
  echo "$N | alone: $A | step: $B" | tee -a $OUT/sweep.txt
done
python -m vacmap_amd.build --force > /dev/null 2>&1
