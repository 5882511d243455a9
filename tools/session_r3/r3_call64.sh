timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "quad or parity" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
for f in 0 1 0 1; do
  echo "== DISYOLO_DGRAD_QUAD=$f"
  DISYOLO_DGRAD_QUAD=$f python bench.py --no-secondary --no-cpu-baseline --no-kernel-events --stage 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stage2', d['value'], d['ms_per_step'], d['config'].get('loss_first'), d['config'].get('loss_last'))"
done
