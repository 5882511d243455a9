timeout 1200 python -m pytest tests/test_gpu_kat.py tests/test_gpu_net.py tests/test_gpu_canary.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -4
for lib in prev hip prev hip; do
  echo "== lib $lib"
  DISYOLO_LIB=$GRAFT_REPO_ROOT/dis-yolo_amd/libdisyolo_$lib.so python bench.py --no-secondary --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage1', d['value'], d['ms_per_step'], d['config'].get('loss_first'), d['config'].get('loss_last'))"
  DISYOLO_LIB=$GRAFT_REPO_ROOT/dis-yolo_amd/libdisyolo_$lib.so python bench.py --no-secondary --no-cpu-baseline --no-kernel-events --stage 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage2', d['value'], d['ms_per_step'], d['config'].get('loss_first'), d['config'].get('loss_last'))"
done
