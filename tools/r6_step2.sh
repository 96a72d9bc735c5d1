mkdir -p gpurun_out/r6c
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hg38.py -m gpu -x -q > gpurun_out/r6c/gputest.log 2>&1; tail -5 gpurun_out/r6c/gputest.log
for i in 1 2; do
timeout 600 python bench.py --extra-configs "" --cpu-sample 0 --no-host-input > gpurun_out/r6c/bench_$i.json 2> gpurun_out/r6c/bench_$i.err
python - <<PY
import json
d=json.load(open('gpurun_out/r6c/bench_$i.json'))
print(round(d['value'],3), round(d['ms_per_step'],2), 'cores', d['host_cores_busy_timed_pass'], 'syncs', d['host_syncs_per_step'], 'stages', [round(x,1) for x in d['stage_ms_per_step']], 'fail', d['failed_reads'], d['oracle_crosscheck'])
PY
done
