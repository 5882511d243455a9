timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "block32 or conv12" -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25
python tools/bench_block32.py 8 288 2>&1 | tail -2
python tools/bench_block32.py 32 288 2>&1 | tail -2
