#!/bin/bash
# which of two bench processes run back to back is the slower one — the first or the second? (main = ONT-hg38 headline config, side = hifi_hg38)
cd $GRAFT_REPO_ROOT
run() { python bench.py --config $1 --steps $2 --warmup 5 --cpu-sample 0 --verify 0 --no-host-input --extra-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  $1:', round(d['ms_per_step'],3), 'ms', round(d['value'],3), 'Gbp/s, seed', round(d['stage_ms_per_step'][0],1))"; }
for rep in 1 2 3; do
  echo "main then side"; run ont_hg38 20; run hifi_hg38 24
  echo "side then main"; run hifi_hg38 24; run ont_hg38 20
done
