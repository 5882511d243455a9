#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_trajectory.py -x -q -s > $O/r06f_traj.txt 2>&1; echo "traj rc=$?"; tail -3 $O/r06f_traj.txt
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_832.py -x -q > $O/r06f_fp8_tests.txt 2>&1; echo "fp8 tests rc=$?"; tail -3 $O/r06f_fp8_tests.txt
timeout 900 python -m pytest tests/test_gpu_dp2.py -x -q > $O/r06f_dp2.txt 2>&1; echo "dp2 rc=$?"; tail -3 $O/r06f_dp2.txt
for mx in 1 0; do
  DISYOLO_FP8_MX=$mx timeout 600 python tools/bench_fp8_layers.py 4 832 > $O/r06f_fp8_layers_832_B4_mx$mx.txt 2>&1; tail -3 $O/r06f_fp8_layers_832_B4_mx$mx.txt
  DISYOLO_FP8_MX=$mx timeout 600 python tools/bench_fp8_layers.py 32 576 > $O/r06f_fp8_layers_576_B32_mx$mx.txt 2>&1; tail -3 $O/r06f_fp8_layers_576_B32_mx$mx.txt
done
