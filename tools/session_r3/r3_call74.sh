R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- python3 $R/bench.py --task infer --batch 32 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary > $O/r03h_infer_under_rocprof.json 2>/dev/null
cp /tmp/pi/*/*kernel_stats.csv $O/r03h_infer_kernel_stats.csv
