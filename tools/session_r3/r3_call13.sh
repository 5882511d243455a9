mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_gpu_832.py tests/test_gpu_train_data.py tests/test_gpu_net.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15
