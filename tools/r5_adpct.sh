#!/bin/bash
# band-width rule of the gap fill (margin >= pct % of the problem size; the proof decides either way): re-tune with six batches in flight
cd $GRAFT_REPO_ROOT
run() { timeout 240 env VMX_AD_PCT=$1 python bench.py --steps 40 --warmup 5 --cpu-sample 0 --verify 16 --no-host-input --extra-configs "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pct $1:', round(d['ms_per_step'],3), 'ms', d['oracle_crosscheck'], 'redo/step', round(d['dp_redo_per_step']), 'cells/read', round(d['per_read']['dp_cells']/1e6,3), 'M, fill+records stage', round(d['stage_ms_per_step'][5],1))"; }
for p in 90 75 60 90 75 60 100; do run $p; done
