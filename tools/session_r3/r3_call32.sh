timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k conv12 2>&1 | tail -5
python tools/bench_conv12.py 8 576 2>&1 | tail -1
python tools/bench_conv12.py 32 576 2>&1 | tail -1
python tools/bench_conv12.py 4 832 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -5
