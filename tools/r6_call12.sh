#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
b() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-box 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'], d['ms_per_step_min_max'], d['config']['loss_last'])
"; }
for r in 1 2; do
b DISYOLO_BN_INKERNEL=0
b DISYOLO_BN_INKERNEL=1 DISYOLO_BN_INKERNEL_BWD=0
b DISYOLO_BN_INKERNEL=1 DISYOLO_BN_INKERNEL_BWD=0 DISYOLO_BN_INKERNEL_FWD_GEMM=0
b DISYOLO_BN_INKERNEL=1 DISYOLO_BN_INKERNEL_BWD=1 DISYOLO_BN_INKERNEL_BWD_GEMM=0
b DISYOLO_BN_INKERNEL=1 DISYOLO_BN_INKERNEL_BWD=1 DISYOLO_BN_INKERNEL_BWD_GEMM=0 DISYOLO_BN_INKERNEL_FWD_GEMM=0
b DISYOLO_BN_INKERNEL=1 DISYOLO_BN_INKERNEL_BWD=1
done
