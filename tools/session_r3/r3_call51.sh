timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -2
for f in 0 1 0 1; do
  echo "== DISYOLO_XCD_N=$f"
  DISYOLO_XCD_N=$f python bench.py --no-secondary --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage1', d['value'], d['ms_per_step'], d['config'].get('loss_last'))"
  DISYOLO_XCD_N=$f python bench.py --no-secondary --no-cpu-baseline --task infer --batch 32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('infer32', d['value'], d['ms_per_step'])"
done
COLD=1 DISYOLO_XCD_N=0 python tools/bench_conv.py 8 576 12,0x20c,3,0x203 2>&1 | grep -E "shape|^\(18, "
COLD=1 DISYOLO_XCD_N=1 python tools/bench_conv.py 8 576 12,0x20c,3,0x203 2>&1 | grep -E "^\(18, "
