for t in profiles/tune_train_B8_576_stage1.json tools/tune_dgrad_gemm.json profiles/tune_train_B8_576_stage1.json tools/tune_dgrad_gemm.json; do
  python bench.py --no-secondary --no-cpu-baseline --no-kernel-events --tune-cache $t 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage1', '$t', d['value'], d['ms_per_step'])"
done
