#!/bin/bash
# round 6, call 1: baseline bench line + SQ counters of the 3x3 conv kernels at the round-5 tree
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/r06_counters_avail.txt 2>&1
python3 $R/tools/conv_counters.py run > $O/r06a_conv_standalone.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/cc1 -- python3 $R/tools/conv_counters.py run > $O/r06a_cc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/cc2 -- python3 $R/tools/conv_counters.py run > $O/r06a_cc2.log 2>&1
python3 $R/tools/conv_counters.py report /tmp/cc1 $O/r06a_conv_sq_counters.txt /tmp/cc2
cd $R
python bench.py --no-cpu-baseline > $O/r06a_bench_stage1.json 2> $O/r06a_bench_err.txt
tail -c 1500 $O/r06a_bench_stage1.json
