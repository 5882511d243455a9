python tools/make_tune_tables.py train1pair 2>&1 | grep -v amdgpu.ids | tail -2
cp gpurun_out/tune_train_B8_576_stage1_pair.json profiles/
python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 > gpurun_out/r03c_bench_with_pair.json
python - <<PY
import json
d=json.loads(open('gpurun_out/r03c_bench_with_pair.json').read())
print('headline', d['value'], d['ms_per_step'])
for k,v in d['secondary'].items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))
PY
