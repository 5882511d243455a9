timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -3
COLD=1 python tools/bench_conv.py 32 576 12,22 2>&1 | grep -v amdgpu.ids | grep -E "shape|, 3, 1\)|, 3, 2\)|, 1, 1\)" | head -30
