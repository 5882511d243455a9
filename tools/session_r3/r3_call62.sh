timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -3
COLD=1 python tools/bench_conv.py 8 576 16,17,18,19,12,0x20c 2>&1 | grep -v amdgpu.ids | grep -E "shape|, 3, 1\)" | head -12
python tools/make_tune_tables.py train1 2>&1 | grep -v amdgpu.ids | tail -1
for t in profiles/tune_train_B8_576_stage1.json gpurun_out/tune_train_B8_576_stage1.json profiles/tune_train_B8_576_stage1.json gpurun_out/tune_train_B8_576_stage1.json; do
  python bench.py --no-secondary --no-cpu-baseline --no-kernel-events --tune-cache $t 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage1', '$t', d['value'], d['ms_per_step'])"
done
python - <<PY
import json
a=json.load(open('profiles/tune_train_B8_576_stage1.json')); b=json.load(open('gpurun_out/tune_train_B8_576_stage1.json'))
for k in b:
    if a.get(k)!=b[k]: print(k, a.get(k), '->', b[k])
PY
