timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k conv12 2>&1 | tail -2
python tools/bench_conv12.py 8 576 2>&1 | tail -1
python tools/bench_conv12.py 32 576 2>&1 | tail -1
python tools/bench_conv12.py 4 832 2>&1 | tail -1
tools/bin/probe_conv12 8 576 | head -5
timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_configs.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -4
