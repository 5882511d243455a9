#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_bn_inkernel.py -x -q > $O/r06c_test_inkernel.txt 2>&1; echo "inkernel rc=$?"
tail -25 $O/r06c_test_inkernel.txt
timeout 1200 python -m pytest tests/test_gpu_net.py -x -q > $O/r06c_test_net.txt 2>&1; echo "net rc=$?"
tail -5 $O/r06c_test_net.txt
for v in "1 1" "1 0" "0 0" "1 1" "0 0"; do
  set -- $v
  DISYOLO_BN_INKERNEL=$1 DISYOLO_BN_INKERNEL_BWD=$2 timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-box 2> $O/r06c_bench_err_$1$2.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('INKERNEL=$1 BWD=$2', d['value'], d['ms_per_step'], d['ms_per_step_min_max'], d['config']['loss_last'])
"
done
