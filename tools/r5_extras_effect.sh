#!/bin/bash
# does running the side configs (own processes, whole HBM each) BEFORE the main run slow the main run's timed pass?
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 20 --warmup 5 --cpu-sample 0 --verify 16 --no-host-input "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*:', round(d['ms_per_step'],3), 'ms, seed stage', round(d['stage_ms_per_step'][0],1), 'host active', round(d['host_active_ms_per_batch'],2), 'setup', round(d['setup_s'],1))"; }
run --extra-configs ""; run --extra-configs hifi_hg38,vacsim_r; run --extra-configs ""; run --extra-configs hifi_hg38,vacsim_r
