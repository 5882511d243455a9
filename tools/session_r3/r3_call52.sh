python tools/make_tune_tables.py train1 train2 infer train1pair 2>&1 | grep -v amdgpu.ids | tail -4
for w in "train_B8_576_stage1:--stage 1" "train_B8_576_stage2:--stage 2" "infer_B32_576:--task infer --batch 32"; do
  name=${w%%:*}; args=${w#*:}
  for t in profiles/tune_$name.json gpurun_out/tune_$name.json; do
    python bench.py --no-secondary --no-cpu-baseline --no-kernel-events $args --tune-cache $t 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', '$t', d['value'], d['ms_per_step'])"
  done
done
python bench.py --no-secondary --no-cpu-baseline --no-kernel-events --pair --tune-cache profiles/tune_train_B8_576_stage1_pair.json 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pair old', d['value'])"
python bench.py --no-secondary --no-cpu-baseline --no-kernel-events --pair --tune-cache gpurun_out/tune_train_B8_576_stage1_pair.json 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pair new', d['value'])"
