mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
timeout 600 python tools/bench_conv.py 8 576 16,17,18,19,12,3 > gpurun_out/r3/bench_conv_b8.txt 2>&1; cat gpurun_out/r3/bench_conv_b8.txt | grep ", 3, 1"
timeout 600 python tools/bench_conv.py 32 576 16,17,18,19,12,3 > gpurun_out/r3/bench_conv_b32.txt 2>&1; cat gpurun_out/r3/bench_conv_b32.txt | grep ", 3, 1"
