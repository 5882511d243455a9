python tools/make_tune_tables.py train2 2>&1 | grep -v amdgpu.ids | tail -1
for t in profiles/tune_train_B8_576_stage2.json gpurun_out/tune_train_B8_576_stage2.json profiles/tune_train_B8_576_stage2.json gpurun_out/tune_train_B8_576_stage2.json; do
  python bench.py --no-secondary --no-cpu-baseline --no-kernel-events --stage 2 --tune-cache $t 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage2', '$t', d['value'], d['ms_per_step'])"
done
python - <<PY
import json
a=json.load(open('profiles/tune_train_B8_576_stage2.json')); b=json.load(open('gpurun_out/tune_train_B8_576_stage2.json'))
for k in b:
    if a.get(k)!=b[k]: print(k, a.get(k), '->', b[k])
PY
