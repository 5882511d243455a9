#!/bin/bash
# round 6, call 2: forward batch norm inside the conv launches -- parity, then the step with it on / off
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_bn_inkernel.py -x -q > $O/r06b_test_inkernel.txt 2>&1; echo "inkernel rc=$?" 
tail -15 $O/r06b_test_inkernel.txt
timeout 1200 python -m pytest tests/test_gpu_net.py -x -q > $O/r06b_test_net.txt 2>&1; echo "net rc=$?"
tail -5 $O/r06b_test_net.txt
for v in 1 0 1 0; do
  DISYOLO_BN_INKERNEL=$v timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-box 2> $O/r06b_bench_err_$v.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('INKERNEL=$v', d['value'], d['ms_per_step'], d['ms_per_step_min_max'], d['config']['loss_last'])
"
done
