python tools/make_tune_tables.py 2>&1 | grep tune_
cp gpurun_out/tune_*.json profiles/
B="python bench.py --no-kernel-events --no-secondary --no-cpu-baseline --steps 20 --warmup 5 --repeats 5"
for tc in auto none auto none; do
  for st in 1 2; do
    $B --stage $st --tune-cache $tc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tc stage', $st, d['value'], d['ms_per_step'], d['config']['loss_last'], d['config']['conv_tiles'][:30])"
  done
  python bench.py --task infer --batch 32 --steps 10 --warmup 3 --repeats 5 --tune-cache $tc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tc infer', d['value'], d['ms_per_step'])"
done
