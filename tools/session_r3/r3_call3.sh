mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_gpu_canary.py -x -q 2>&1 | tail -15 > gpurun_out/r3/canary.log; cat gpurun_out/r3/canary.log
for i in 1 2; do
  timeout 300 python tools/nan_bisect.py --stage 2 --steps 70 --tag "H$i" > gpurun_out/r3/bisect_H$i.log 2>&1; tail -1 gpurun_out/r3/bisect_H$i.log
done
timeout 300 python tools/nan_bisect.py --stage 1 --steps 70 --tag "S1" > gpurun_out/r3/bisect_S1.log 2>&1; tail -1 gpurun_out/r3/bisect_S1.log
