#!/bin/bash
# On the GPU box: the short evidence refresh on the last commit of round 6 (GPU suite, default bench line, the driver's command)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6zz
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_zzz_gputest_last_commit.log 2>&1; tail -2 gpurun_out/r06_zzz_gputest_last_commit.log
timeout 1500 python bench.py > gpurun_out/r06_zzz_bench_line_default_last_commit.json 2> gpurun_out/r6zz/bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_zzz_bench_line_driver_command_last_commit.json 2>> gpurun_out/r6zz/bench.err
python - <<P
import json
for f in ('r06_zzz_bench_line_default_last_commit', 'r06_zzz_bench_line_driver_command_last_commit'):
    d = json.load(open('gpurun_out/%s.json' % f))
    print(f, round(d['value'], 3), round(d['ms_per_step'], 2), 'syncs', d['host_syncs_per_step'], 'cores', d['host_cores_busy_timed_pass'], d['oracle_crosscheck'], [(e['config'], round(e['value'], 3), e['oracle_crosscheck']) for e in d.get('extra', {}).get('configs', [])])
P
