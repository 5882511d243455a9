timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_net.py tests/test_gpu_fullsize.py tests/test_gpu_832.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for f in 0 1 0 1; do
  echo "== DISYOLO_FUSE_B64=$f"
  DISYOLO_FUSE_B64=$f python bench.py --no-secondary --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage1', d['value'], d['ms_per_step'], d['config'].get('loss_last'))"
  DISYOLO_FUSE_B64=$f python bench.py --no-secondary --no-cpu-baseline --task infer --batch 32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('infer32', d['value'], d['ms_per_step'])"
done
