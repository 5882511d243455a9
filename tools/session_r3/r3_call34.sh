timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k conv12 2>&1 | tail -3
python tools/bench_conv12.py 8 576 2>&1 | tail -1
python tools/bench_conv12.py 32 576 2>&1 | tail -1
tools/bin/probe_conv12 8 576 | head -5
for f in 0 1; do
  echo "== DISYOLO_FUSE12=$f"
  DISYOLO_FUSE12=$f python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage1', d['value'], d['ms_per_step'], d.get('loss_first'), d.get('loss_last'))"
  DISYOLO_FUSE12=$f python bench.py --no-secondary --task infer --batch 32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('infer32', d['value'], d['ms_per_step'])"
done
timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -4
