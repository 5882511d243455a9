timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "batchnorm_backward or partials" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15
for f in 0 1; do
  echo "== DISYOLO_BN_FUSE_GEMM=$f"
  DISYOLO_BN_FUSE_GEMM=$f python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage1', d['value'], d['ms_per_step'], d['config'].get('loss_first'), d['config'].get('loss_last'))"
  DISYOLO_BN_FUSE_GEMM=$f python bench.py --no-secondary --no-cpu-baseline --stage 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage2', d['value'], d['ms_per_step'], d['config'].get('loss_first'), d['config'].get('loss_last'))"
done
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_canary.py tests/test_gpu_configs.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert " | tail -8
