python tools/bench_conv12.py 8 576 2>&1 | tail -1
python tools/bench_conv12.py 32 576 2>&1 | tail -1
python tools/bench_conv12.py 4 832 2>&1 | tail -1
timeout 2200 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25
