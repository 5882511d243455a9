timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "block64" -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
python tools/bench_block32.py 8 288 2>&1 | tail -1
python tools/bench_block32.py 32 288 2>&1 | tail -1
tools/bin/probe_block64 32 144 | head -6 | cut -c1-330
