#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
run() { echo "=== $*"; env "$@" timeout 120 python tools/bn_inkernel_debug.py 8 $MODE 2>&1 | grep -v amdgpu.ids | head -12; }
MODE=joined_nosync run A=1
MODE=overlap run DISYOLO_WG3_BLOCKS=64 DISYOLO_WG_BLOCKS=64
MODE=overlap run DISYOLO_LANE1_LOW=0
MODE=overlap run DISYOLO_BN_INKERNEL_BWD_GEMM=0
MODE=overlap run DISYOLO_OPT_OVERLAP=0
