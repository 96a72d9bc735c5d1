#!/bin/bash
# On the GPU box: the default library against variants (VACMAPX_LIB) on one bench workload, alternating runs. Usage: tools/ab_cfg.sh <config> <rounds> lib1.so [lib2.so ...]
cd $GRAFT_REPO_ROOT
CFG=$1; N=$2; shift; shift
for i in $(seq 1 $N); do
  for L in default "$@"; do
    if [ $L = default ]; then unset VACMAPX_LIB; else export VACMAPX_LIB=$PWD/$L; fi
    timeout 500 python bench.py --config $CFG --extra-configs "" --cpu-sample 0 --verify 16 --no-host-input --steps 24 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$CFG', '$L', round(d['value'], 3), 'Gbp/s', round(d['ms_per_step'], 2), 'ms/step', d['oracle_crosscheck'], 'band ms', round(d['roofline']['kernels']['k_local_seed']['ms_per_step'], 2), 'general', d['local_general_reads'])"
  done
done
