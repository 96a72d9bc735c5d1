mkdir -p gpurun_out/r6b
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/r6b/gputest_kernels.log 2>&1; tail -3 gpurun_out/r6b/gputest_kernels.log
for mode in sleep spin sleep spin; do
VMX_WAIT_MODE=$mode timeout 600 python bench.py --extra-configs "" --cpu-sample 0 --no-host-input > gpurun_out/r6b/bench_$mode.json 2> gpurun_out/r6b/bench_$mode.err
python - <<PY
import json
d=json.load(open('gpurun_out/r6b/bench_$mode.json'))
print('$mode', round(d['value'],3), round(d['ms_per_step'],2), 'cores', d['host_cores_busy_timed_pass'], 'syncs', d['host_syncs_per_step'], 'wait', round(d['host_wait_ms_per_batch'],1), 'active', round(d['host_active_ms_per_batch'],1))
PY
done
