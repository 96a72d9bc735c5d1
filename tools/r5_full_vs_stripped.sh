#!/bin/bash
# does anything the default command does BEFORE its timed pass (the oracle cross-check of 64 reads: index build + alignments on the host) slow the timed pass down?
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 20 --warmup 5 --extra-configs "" --cpu-sample 0 --no-host-input "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*:', round(d['ms_per_step'],3), 'ms', d['oracle_crosscheck'], 'host active', round(d['host_active_ms_per_batch'],2), 'wait', round(d['host_wait_ms_per_batch'],1))"; }
run --verify 64; run --verify 0; run --verify 64; run --verify 0; run --verify 16
