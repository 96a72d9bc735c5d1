#!/bin/bash
# a bench process that starts right after another one has used and freed the whole HBM runs slower: does waiting help?
cd $GRAFT_REPO_ROOT
run() { python bench.py --config vacsim_r --steps 24 --warmup 5 --cpu-sample 0 --verify 0 --no-host-input --extra-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1:', round(d['ms_per_step'],3), 'ms', round(d['value'],3), 'Gbp/s, seed', round(d['stage_ms_per_step'][0],1), 'local', round(d['stage_ms_per_step'][2],1), 'fill', round(d['stage_ms_per_step'][5],1), 'setup', round(d['setup_s'],1), 'hbm', round(d['hbm_used_gb'],1))"; }
run fresh; run right_after; sleep 30; run after_30s; run right_after_again
