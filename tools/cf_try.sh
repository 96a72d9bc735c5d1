#!/bin/bash
# On the GPU box: k_cluster_big phase ticks (library built with -DVMX_CF_TICKS) + one-batch-in-flight kernel summary + seed GPU tests.
# Usage: tools/cf_try.sh <tag>
TAG=${1:-cf}
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --cpu-sample 0 --verify 64 --streams 1 --steps 6 --extra-configs "" > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err
grep "cf ticks" gpurun_out/$TAG.err | tail -1
python - <<P
import json
d=json.loads(open('gpurun_out/$TAG.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'xcheck', d.get('extra',{}).get('cross_check', d.get('cross_check')))
P
timeout 300 python -m pytest tests -x -q -m gpu -k "seed" 2>&1 | tail -2
bash tools/prof1.sh $TAG --extra-configs "" 2>&1 | grep -i "cluster\|sketch\|fill_hits\|total\|lookup" | head
