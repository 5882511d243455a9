timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "stream" 2>&1 | grep -E "Error|error|assert|check|mismatch|^E " | head -30
