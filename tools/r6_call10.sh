#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_bn_inkernel.py -x -q > $O/r06d_test_inkernel.txt 2>&1; echo "inkernel rc=$?"
tail -12 $O/r06d_test_inkernel.txt
for m in overlap joined_nosync; do echo "== $m"; timeout 120 python tools/bn_inkernel_debug.py 8 $m 2>&1 | grep -v amdgpu.ids | head -8; done
for v in "1 1" "1 0" "0 0" "1 1" "0 0"; do
  set -- $v
  DISYOLO_BN_INKERNEL=$1 DISYOLO_BN_INKERNEL_BWD=$2 timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-box 2> $O/r06d_bench_err_$1$2.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('INKERNEL=$1 BWD=$2', d['value'], d['ms_per_step'], d['ms_per_step_min_max'], d['config']['loss_last'])
"
done
