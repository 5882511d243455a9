timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "fused_launches" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
