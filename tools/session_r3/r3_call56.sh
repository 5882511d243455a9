R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps3 -- python3 $R/bench.py --stage 2 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary --no-kernel-events > $O/r03e_bench_stage2_under_rocprof.json 2>/dev/null
cp /tmp/ps3/*/*kernel_stats.csv $O/r03e_bench_stage2_kernel_stats.csv
