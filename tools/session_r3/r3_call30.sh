timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "conv12" 2>&1 | tail -12
