#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_832.py tests/test_gpu_conv.py tests/test_gpu_fullsize.py -x -q > $O/r06g_tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/r06g_tests.txt
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'], d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'))
"; }
C="--steps 10 --warmup 3 --repeats 5 --no-secondary --no-cpu-baseline --no-kernel-events --no-box"
for r in 1 2; do
echo "== 832 B4 train"; b timeout 300 python bench.py --size 832 --batch 4 $C --dtype bf16; b timeout 300 python bench.py --size 832 --batch 4 $C --dtype fp8; b DISYOLO_FP8_FROM=1 timeout 300 python bench.py --size 832 --batch 4 $C --dtype fp8
echo "== infer B32"; b timeout 300 python bench.py --task infer --batch 32 $C --dtype bf16; b timeout 300 python bench.py --task infer --batch 32 $C --dtype fp8
echo "== halo split rows 1/0"; b DISYOLO_HALO_SPLIT_ROWS=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-box; b DISYOLO_HALO_SPLIT_ROWS=0 timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-box
done
for v in 1 0; do DISYOLO_HALO_SPLIT_ROWS=$v CC_ONLY=halo timeout 200 python tools/conv_counters.py run 2>&1 | grep CASE; done
