mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3/gpu_suite1.log; cat gpurun_out/r3/gpu_suite1.log
timeout 900 python bench.py > gpurun_out/r3/bench1.json 2> gpurun_out/r3/bench1.err; echo rc=$?; tail -3 gpurun_out/r3/bench1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_min_max')}, d['config'].get('loss_first'), d['config'].get('loss_last'))
print(json.dumps(d.get('secondary'))[:1500])
print(d.get('roofline'))
PY
