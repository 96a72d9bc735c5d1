#!/bin/bash
# On the GPU box: the default bench (no side configs, no CPU leg) under a list of environment settings, one line per setting.
# Usage: tools/envsweep.sh "A=1 B=2" "A=3" ...      ("" = defaults)
cd $GRAFT_REPO_ROOT
for S in "$@"; do
  env $S timeout 400 python bench.py --extra-configs "" --cpu-sample 0 --verify 0 > /tmp/es.json 2>/dev/null
  python - "$S" <<P
import json,sys
try:
    d=json.loads(open('/tmp/es.json').read().strip().splitlines()[-1]); print('%-40s ms_per_step %.2f  value %.3f' % (sys.argv[1] or '(default)', d['ms_per_step'], d['value']))
except Exception as e: print(sys.argv[1], 'FAILED', e)
P
done
