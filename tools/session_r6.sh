#!/bin/bash
# Round 6: the gpurun calls of the round, one shell function per call (run on the GPU box from the repository root:
#   gpurun -- "bash tools/session_r6.sh callN").  Outputs under gpurun_out/; what was kept is in profiles/r06_*.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R

call1() {
  # round 6, call 1: baseline bench line + SQ counters of the 3x3 conv kernels at the round-5 tree
  cd /tmp; export TMPDIR=/tmp
  rocprofv3 -L > $O/r06_counters_avail.txt 2>&1
  python3 $R/tools/conv_counters.py run > $O/r06a_conv_standalone.txt 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/cc1 -- python3 $R/tools/conv_counters.py run > $O/r06a_cc1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/cc2 -- python3 $R/tools/conv_counters.py run > $O/r06a_cc2.log 2>&1
  python3 $R/tools/conv_counters.py report /tmp/cc1 $O/r06a_conv_sq_counters.txt /tmp/cc2
  python bench.py --no-cpu-baseline > $O/r06a_bench_stage1.json 2> $O/r06a_bench_err.txt
  tail -c 1500 $O/r06a_bench_stage1.json
}

call2() {
  # round 6, call 2: forward batch norm inside the conv launches -- parity, then the step with it on / off
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
}

call5() {
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
}

call9() {
  run() { echo "=== $*"; env "$@" timeout 120 python tools/bn_inkernel_debug.py 8 $MODE 2>&1 | grep -v amdgpu.ids | head -12; }
  MODE=joined_nosync run A=1
  MODE=overlap run DISYOLO_WG3_BLOCKS=64 DISYOLO_WG_BLOCKS=64
  MODE=overlap run DISYOLO_LANE1_LOW=0
  MODE=overlap run DISYOLO_BN_INKERNEL_BWD_GEMM=0
  MODE=overlap run DISYOLO_OPT_OVERLAP=0
}

call10() {
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
}

call12() {
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
}

call15() {
  timeout 900 python -m pytest tests/test_gpu_trajectory.py -x -q -s > $O/r06f_traj.txt 2>&1; echo "traj rc=$?"; tail -3 $O/r06f_traj.txt
  timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_832.py -x -q > $O/r06f_fp8_tests.txt 2>&1; echo "fp8 tests rc=$?"; tail -3 $O/r06f_fp8_tests.txt
  timeout 900 python -m pytest tests/test_gpu_dp2.py -x -q > $O/r06f_dp2.txt 2>&1; echo "dp2 rc=$?"; tail -3 $O/r06f_dp2.txt
  for mx in 1 0; do
    DISYOLO_FP8_MX=$mx timeout 600 python tools/bench_fp8_layers.py 4 832 > $O/r06f_fp8_layers_832_B4_mx$mx.txt 2>&1; tail -3 $O/r06f_fp8_layers_832_B4_mx$mx.txt
    DISYOLO_FP8_MX=$mx timeout 600 python tools/bench_fp8_layers.py 32 576 > $O/r06f_fp8_layers_576_B32_mx$mx.txt 2>&1; tail -3 $O/r06f_fp8_layers_576_B32_mx$mx.txt
  done
}

call16() {
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
}

call17() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:70], '|', d['config']['workload'], d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('bn_inkernel',{}).get('forward_layers'))
"; }
C="--no-secondary --no-cpu-baseline --no-box"
for r in 1 2; do
b DISYOLO_BN_INKERNEL_ROW_KB=128 timeout 300 python bench.py $C
b DISYOLO_BN_INKERNEL_ROW_KB=256 timeout 300 python bench.py $C
b DISYOLO_BN_INKERNEL=0 timeout 300 python bench.py $C
b DISYOLO_BN_INKERNEL=1 timeout 300 python bench.py $C --stage 2 --steps 10 --repeats 5
b DISYOLO_BN_INKERNEL=0 timeout 300 python bench.py $C --stage 2 --steps 10 --repeats 5
b DISYOLO_X=1 timeout 300 python bench.py $C --dtype fp8
done
}

call22() {
timeout 900 python tools/instep_sweep.py --stage 2 --steps 20 --table profiles/tune_train_B8_576_stage2.json --compare profiles/tune_train_B8_576_stage2_r06cand.json --rounds 4 > $O/r06_sweep2_compare.txt 2>&1; tail -3 $O/r06_sweep2_compare.txt | cut -c1-300
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_step
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_step -- python3 $R/bench.py --no-secondary --no-cpu-baseline --no-box --no-kernel-events --steps 10 --repeats 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, re, collections, os
f = glob.glob('/tmp/pmc_step/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|void |dyconv::|HIP_vector_type<[^>]*>|\(.*$", "", r["Kernel_Name"])
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[name].add(r["Dispatch_Id"])
out = open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_step_sq_counters.txt", "w")
hdr = "%-52s %6s %11s %7s %7s %7s %9s %9s %9s" % ("kernel", "launch", "wavecyc/l", "parked", "stalled", "issuing", "mfma/wave", "lds_act/w", "lds_cnfl/w")
print(hdr); out.write(hdr + "\n")
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:30]:
    k = len(n[name]); wc = c["SQ_WAVE_CYCLES"] or 1
    line = "%-52s %6d %11.0f %6.1f%% %6.1f%% %6.1f%% %8.2f%% %8.2f%% %8.2f%%" % (name[:52], k, wc / k, 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc,
        100 * c["SQ_ACTIVE_INST_ANY"] / wc, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc), 100 * c["SQ_LDS_IDX_ACTIVE"] / (4 * wc), 100 * c["SQ_LDS_BANK_CONFLICT"] / (4 * wc))
    print(line); out.write(line + "\n")
PY
}

call28() {
timeout 1200 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_train_data.py tests/test_gpu_net.py -x -q > $O/r06k_tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/r06k_tests.txt
b() { "$@" 2>$O/r06k_err.txt | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$*'[-60:], '|', d['config']['workload'], d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), 'pipe' if d['config'].get('backbone_pipeline') else 'plain', d['config'].get('loss_last'))
except Exception as e:
    print('$*', 'FAILED', e); print(open('$O/r06k_err.txt').read()[-1500:])
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
b timeout 300 python bench.py $C
b timeout 300 python bench.py $C --pipeline off
b timeout 300 python bench.py $C --feed per-step
b timeout 300 python bench.py $C --feed per-step --pipeline off
done
b timeout 300 python bench.py $C --dtype fp8
b timeout 300 python bench.py $C --dtype fp8 --pipeline off
b timeout 300 python bench.py $C --size 832 --batch 4 --steps 10 --repeats 5
b timeout 300 python bench.py $C --size 832 --batch 4 --steps 10 --repeats 5 --dtype fp8
b timeout 300 python bench.py $C --force-dp
}

call29() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:58], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
b A=1 timeout 300 python bench.py $C
b DISYOLO_LANE2_LOW=1 timeout 300 python bench.py $C
b DISYOLO_BN_INKERNEL=0 timeout 300 python bench.py $C
b DISYOLO_WG3_BLOCKS=160 DISYOLO_WG_BLOCKS=160 timeout 300 python bench.py $C
b DISYOLO_WG3_BLOCKS=224 DISYOLO_WG_BLOCKS=224 timeout 300 python bench.py $C
b DISYOLO_WG3_BLOCKS=256 DISYOLO_WG_BLOCKS=256 timeout 300 python bench.py $C
b DISYOLO_WGRAD_GROUP=2 timeout 300 python bench.py $C
b DISYOLO_WGRAD_GROUP=4 timeout 300 python bench.py $C
b DISYOLO_HALO_SPLIT_ROWS=0 timeout 300 python bench.py $C
done
}

call30() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:64], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2 3; do
b DISYOLO_LANE2_LOW=1 timeout 300 python bench.py $C
b DISYOLO_LANE2_LOW=1 DISYOLO_WGRAD_GROUP=2 timeout 300 python bench.py $C
b DISYOLO_LANE2_LOW=1 DISYOLO_WGRAD_GROUP=1 timeout 300 python bench.py $C
b DISYOLO_LANE2_LOW=1 DISYOLO_LANE1_LOW=0 timeout 300 python bench.py $C
b DISYOLO_LANE2_LOW=1 DISYOLO_BN_INKERNEL=0 timeout 300 python bench.py $C
b DISYOLO_LANE2_LOW=1 DISYOLO_TAIL_MAIN=2 timeout 300 python bench.py $C
done
}

call31() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:60], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2 3; do
for t in 0 1 2 3 4 6 8; do
b DISYOLO_LANE2_LOW=1 DISYOLO_TAIL_MAIN=$t timeout 300 python bench.py $C
done
done
}

call32() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:84], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2 3; do
b DISYOLO_LANE2_LOW=1 DISYOLO_TAIL_MAIN=2 timeout 300 python bench.py $C
b DISYOLO_LANE2_LOW=1 DISYOLO_TAIL_MAIN=2 DISYOLO_PIPE_EARLY=1 timeout 300 python bench.py $C
b DISYOLO_LANE2_LOW=1 DISYOLO_TAIL_MAIN=3 DISYOLO_PIPE_EARLY=1 timeout 300 python bench.py $C
b DISYOLO_TAIL_MAIN=2 DISYOLO_PIPE_EARLY=1 timeout 300 python bench.py $C
done
}

call33() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:70], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2 3; do
b A=1 timeout 300 python bench.py $C
b DISYOLO_TAIL_MAIN=0 timeout 300 python bench.py $C
b A=1 timeout 300 python bench.py $C --pipeline off
done
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_drivers.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
}

call35() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:70], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
b A=1 timeout 300 python bench.py $C
b DISYOLO_BN_INKERNEL_BWD=1 timeout 300 python bench.py $C
b DISYOLO_BN_INKERNEL_ROW_KB=256 timeout 300 python bench.py $C
b DISYOLO_OPT_CHUNK_M=4 timeout 300 python bench.py $C
b DISYOLO_OPT_CHUNK_M=16 timeout 300 python bench.py $C
done
timeout 1500 python tools/instep_sweep.py --stage 1 --out gpurun_out/r06_sweep1p.json --budget 1000 > $O/r06_sweep1p.txt 2>&1; grep -E "^base|^final" $O/r06_sweep1p.txt
}

call36() {
timeout 900 python tools/instep_sweep.py --stage 1 --table profiles/tune_train_B8_576_stage1.json --compare profiles/tune_train_B8_576_stage1_r06p.json --rounds 6 > $O/r06_sweep1p_compare.txt 2>&1; tail -3 $O/r06_sweep1p_compare.txt | cut -c1-300
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:70], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
for m in 8 12 16 24 32; do b DISYOLO_OPT_CHUNK_M=$m timeout 300 python bench.py $C; done
b DISYOLO_OPT_CHUNK_M=16 timeout 300 python bench.py $C --tune-cache profiles/tune_train_B8_576_stage1_r06p.json
done
}

call37() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:60], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
b A=1 timeout 300 python bench.py $C
b DISYOLO_OPT_OVERLAP=1 timeout 300 python bench.py $C
b DISYOLO_BN_INKERNEL=0 timeout 300 python bench.py $C
b DISYOLO_WGRAD_GROUP=1 timeout 300 python bench.py $C
b DISYOLO_WGRAD_GROUP=2 timeout 300 python bench.py $C
b DISYOLO_WGRAD_GROUP=4 timeout 300 python bench.py $C
b DISYOLO_WGRAD_GROUP=6 timeout 300 python bench.py $C
b DISYOLO_LANE1_LOW=0 timeout 300 python bench.py $C
b DISYOLO_TAIL_MAIN=3 timeout 300 python bench.py $C
b DISYOLO_TAIL_MAIN=1 timeout 300 python bench.py $C
done
}

call38() {
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "batchnorm_backward or refuse_the_flag or partials" > $O/r06_t38a.txt 2>&1; echo "conv rc=$?"; tail -5 $O/r06_t38a.txt
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_bn_inkernel.py -x -q > $O/r06_t38b.txt 2>&1; echo "net rc=$?"; tail -5 $O/r06_t38b.txt
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:60], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2 3; do
b DISYOLO_BN_BWD_STATS_GEMM=1 timeout 300 python bench.py $C
b DISYOLO_BN_BWD_STATS_GEMM=0 timeout 300 python bench.py $C
done
for r in 1 2; do
b DISYOLO_BN_BWD_STATS_GEMM=1 timeout 300 python bench.py $C --stage 2 --steps 10 --repeats 5
b DISYOLO_BN_BWD_STATS_GEMM=0 timeout 300 python bench.py $C --stage 2 --steps 10 --repeats 5
done
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import disyolo_amd
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch
for stage in (1, 2):
    net = YOLONet(training=True, device=torch.device("cuda:0"), image_size=576, batch_size=8, stage=stage, seed=0)
    net.set_batch(synthetic_batch(8, 576, seed=1))
    net.autotune(cache="profiles/tune_train_B8_576_stage%d.json" % stage)
    net.train_step(); torch.cuda.synchronize()
    tr = [l for l in net.layers if not l.lock and l.kind != "lin"]
    print("stage", stage, "trainable BN layers", len(tr), "with partial rows", sum(1 for l in tr if l.bwd_part_rows), "without:", [l.idx for l in tr if not l.bwd_part_rows])
PY
}

call40() {
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_configs.py tests/test_gpu_drivers.py tests/test_gpu_conv.py -x -q > $O/r06_t40.txt 2>&1; echo "tests rc=$?"; tail -5 $O/r06_t40.txt
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:80], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2 3; do
b A=1 timeout 300 python bench.py $C
b A=1 timeout 300 python bench.py $C --feed per-step
done
}

call41() {
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_configs.py tests/test_gpu_drivers.py -x -q > $O/r06_t41.txt 2>&1; echo "tests rc=$?"; tail -5 $O/r06_t41.txt
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:80], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2 3; do
b A=1 timeout 300 python bench.py $C
b DISYOLO_FEED_LOW=1 timeout 300 python bench.py $C --feed per-step
b DISYOLO_FEED_LOW=0 timeout 300 python bench.py $C --feed per-step
done
}

call42() {
timeout 2400 python -m pytest tests/test_gpu_net.py tests/test_gpu_configs.py tests/test_gpu_drivers.py tests/test_gpu_train_data.py tests/test_gpu_dp2.py -x -q > $O/r06_t42.txt 2>&1; echo "tests rc=$?"; tail -5 $O/r06_t42.txt
timeout 600 python examples/train_synthetic.py > $O/r06_example.txt 2>&1; echo "example rc=$?"; tail -5 $O/r06_example.txt
}

call45() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:60], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
for n in 0 3 6 9 12 18; do b DISYOLO_PIPE_AFTER=$n timeout 300 python bench.py $C; done
done
}

call46() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:60], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
for n in 0 55 57 61 65 69 73 77 81; do b DISYOLO_PIPE_FWD_AFTER=$n timeout 300 python bench.py $C; done
done
}

call47() {
for r in 1 2; do
timeout 600 python tools/solver_rate.py 2>&1 | tail -1
DISYOLO_FEED_STREAM=0 timeout 600 python tools/solver_rate.py 2>&1 | tail -1
timeout 600 python tools/solver_rate.py --pipeline off 2>&1 | tail -1
done
}

call48() {
timeout 900 python -m pytest tests/test_gpu_train_data.py tests/test_gpu_drivers.py -x -q > $O/r06_t48.txt 2>&1; echo "tests rc=$?"; tail -4 $O/r06_t48.txt
for r in 1 2; do
timeout 600 python tools/solver_rate.py 2>&1 | tail -1
DISYOLO_FEED_STREAM=0 timeout 600 python tools/solver_rate.py 2>&1 | tail -1
done
timeout 600 python tools/solver_rate.py --pipeline off 2>&1 | tail -1
}

call49() {
timeout 900 python -m pytest tests/test_gpu_train_data.py tests/test_gpu_drivers.py -x -q > $O/r06_t49.txt 2>&1; echo "tests rc=$?"; tail -4 $O/r06_t49.txt
for r in 1 2; do
timeout 600 python tools/solver_rate.py 2>&1 | tail -1
SOLVER_RATE_HOST_LABELS=1 timeout 600 python tools/solver_rate.py 2>&1 | tail -1
done
timeout 600 python tools/solver_rate.py --pipeline off 2>&1 | tail -1
}

call51() {
timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/r06_full_suite.txt 2>&1; echo "suite rc=$?"; tail -4 $O/r06_full_suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
}

call52() {
timeout 900 python -m pytest tests/test_gpu_drivers.py -x -q > $O/r06_t52.txt 2>&1; echo "tests rc=$?"; tail -4 $O/r06_t52.txt
timeout 600 python tools/feed_rate.py 1 2>&1 | tail -3
timeout 600 python tools/feed_rate.py 1 plain 2>&1 | tail -3
timeout 600 python tools/feed_rate.py 2 2>&1 | tail -3
}

call53() {
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_drivers.py tests/test_gpu_train_data.py tests/test_gpu_configs.py -x -q > $O/r06_t53.txt 2>&1; echo "tests rc=$?"; tail -4 $O/r06_t53.txt
timeout 600 python tools/feed_rate.py 1 2>&1 | tail -3
DISYOLO_FEED_LANE=0 timeout 600 python tools/feed_rate.py 1 2>&1 | tail -3
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:80], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
b A=1 timeout 300 python bench.py $C
b DISYOLO_FEED_LANE=1 timeout 300 python bench.py $C --feed per-step
b DISYOLO_FEED_LANE=0 timeout 300 python bench.py $C --feed per-step
timeout 600 python tools/solver_rate.py 2>&1 | tail -1
DISYOLO_FEED_LANE=0 timeout 600 python tools/solver_rate.py 2>&1 | tail -1
done
}

call54() {
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dp2.py tests/test_gpu_net.py -x -q > $O/r06_t54.txt 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/r06_t54.txt | tail -2
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:80], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
b A=1 timeout 300 python bench.py $C
b DISYOLO_DP_TAIL_MAIN=1 timeout 300 python bench.py $C --force-dp
b DISYOLO_DP_TAIL_MAIN=0 timeout 300 python bench.py $C --force-dp
done
}

call55() {
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps2 -- python3 $R/bench.py --stage 2 --steps 10 --repeats 3 --no-cpu-baseline --no-secondary --no-box --no-kernel-events > $O/r06_stage2_under_rocprof.json 2>/dev/null
cp /tmp/ps2/*/*kernel_stats.csv $O/r06_stage2_kernel_stats.csv
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:80], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
b DISYOLO_BN_INKERNEL=0 timeout 300 python bench.py $C
b DISYOLO_BN_INKERNEL=0 timeout 300 python bench.py $C --force-dp
}

call57() {
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q > $O/r06_t57.txt 2>&1; echo "conv tests rc=$?"; tail -3 $O/r06_t57.txt
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('$*'[:70], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'), {n[:44]: v['avg_us'] for n, v in list(k.items())[:3]})
"; }
C="--no-secondary --no-cpu-baseline --no-box"
for r in 1 2; do
b A=1 timeout 300 python bench.py $C
b A=1 timeout 300 python bench.py $C --stage 2 --steps 10 --repeats 5
b A=1 timeout 300 python bench.py $C --task infer --batch 32 --steps 10 --repeats 5
done
}

call58() {
for shape in "8 18 512 1024 3" "8 18 1024 512 3" "8 36 256 512 3" "8 18 1024 512 1" "8 36 512 256 1" "32 18 512 1024 3" "32 36 256 512 3"; do
  echo "== $shape  (rotated | straight)"
  timeout 200 python tools/ab_tile.py $shape 12 0x20c 2>&1 | tail -2
  DISYOLO_LIB=$R/dis-yolo_amd/libdisyolo_straight.so timeout 200 python tools/ab_tile.py $shape 12 0x20c 2>&1 | tail -2
done
}

call59() {

for r in 1 2; do
timeout 600 python tools/solver_rate.py --stage 2 --steps 150 2>&1 | tail -1
DISYOLO_SOLVER_AHEAD=0 timeout 600 python tools/solver_rate.py --stage 2 --steps 150 2>&1 | tail -1
timeout 600 python tools/solver_rate.py --pipeline off 2>&1 | tail -1
DISYOLO_SOLVER_AHEAD=0 timeout 600 python tools/solver_rate.py --pipeline off 2>&1 | tail -1
done
}

call62() {
timeout 900 python tools/solver_rate.py --steps 4000 2>&1 | tail -2
timeout 900 python tools/solver_rate.py --steps 1500 --stage 2 2>&1 | tail -2
timeout 600 python bench.py --steps 300 --repeats 3 --no-secondary --no-cpu-baseline --no-box --no-kernel-events --feed per-step 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 300-step regions, fed:', d['value'], d['ms_per_step'], d['ms_per_step_min_max'], d['config']['loss_last'], d['config']['steps_trained'])"
}

call63() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:90], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'), d['config'].get('loss_last'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events"
for r in 1 2; do
for m in 8 32; do
b DISYOLO_OPT_CHUNK_M=$m timeout 300 python bench.py $C
b DISYOLO_OPT_CHUNK_M=$m timeout 300 python bench.py $C --pipeline off
b DISYOLO_OPT_CHUNK_M=$m timeout 300 python bench.py $C --stage 2 --steps 10 --repeats 5
b DISYOLO_OPT_CHUNK_M=$m timeout 300 python bench.py $C --force-dp
done
done
}

call64() {
timeout 900 python -m pytest tests/test_gpu_drivers.py tests/test_gpu_e2e_parity.py -x -q > $O/r06_t64.txt 2>&1; echo "tests rc=$?"; tail -3 $O/r06_t64.txt
for r in 1 2; do
timeout 600 python tools/evaluate_rate.py 2>&1 | tail -1
DISYOLO_EVAL_REPLAY=0 timeout 600 python tools/evaluate_rate.py 2>&1 | tail -1
done
}

call66() {
timeout 900 python -m pytest tests/test_gpu_drivers.py tests/test_voc_eval.py tests/test_postprocess.py -x -q > $O/r06_t66.txt 2>&1; echo "tests rc=$?"; tail -3 $O/r06_t66.txt
for r in 1 2; do
timeout 600 python tools/evaluate_rate.py 2>&1 | tail -1
DISYOLO_EVAL_GPU_IOU=0 timeout 600 python tools/evaluate_rate.py 2>&1 | tail -1
done
}

call69() {
b() { env "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*'[:100], '|', d['value'], d['ms_per_step'], d.get('ms_per_step_min_max'))
"; }
C="--no-secondary --no-cpu-baseline --no-box --no-kernel-events --stage 2 --steps 10 --repeats 5"
for r in 1 2; do
b A=1 timeout 300 python bench.py $C
b A=1 timeout 300 python bench.py $C --feed per-step
b A=1 timeout 300 python bench.py $C --overlap-tail off
done
timeout 300 python tools/host_enqueue.py 2 2>&1 | tail -3
}

call70() {
for r in 1 2; do
DISYOLO_SOLVER_AHEAD=1 timeout 600 python tools/solver_rate.py --stage 2 --steps 150 2>&1 | tail -1
DISYOLO_SOLVER_AHEAD=0 timeout 600 python tools/solver_rate.py --stage 2 --steps 150 2>&1 | tail -1
done
}

call71() {
timeout 900 python -m pytest tests/test_gpu_train_data.py tests/test_gpu_drivers.py -x -q > $O/r06_t71.txt 2>&1; echo "tests rc=$?"; tail -5 $O/r06_t71.txt
for r in 1 2; do
timeout 600 python tools/solver_rate.py 2>&1 | tail -1
timeout 600 python tools/solver_rate.py --stage 2 --steps 150 2>&1 | tail -1
done
}

"$@"
