R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps4 -- python3 $R/bench.py --stage 2 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary --no-kernel-events > /dev/null 2>&1
cp /tmp/ps4/*/*kernel_stats.csv $O/r03g_bench_stage2_kernel_stats.csv
