R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps2 -- python3 $R/bench.py --stage 2 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary --no-kernel-events > $O/r03d_bench_stage2_under_rocprof.json 2>/dev/null
cp /tmp/ps2/*/*kernel_stats.csv $O/r03d_bench_stage2_kernel_stats.csv
cd $R
python tools/make_tune_tables.py train832 train832fp8 2>&1 | grep -v amdgpu.ids | tail -2
