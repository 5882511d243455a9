timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_canary.py tests/test_gpu_net.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
bash tools/r3_call12.sh
