timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
timeout 600 python tools/bench_conv.py 8 576 16,22,12,18 2>&1 | grep ", 3, 1"
timeout 600 python tools/bench_conv.py 32 576 16,22,12,18 2>&1 | grep ", 3, 1"
