# one gpurun call: bench line + rocprofv3 kernel stats + PMC traffic for the SAME tuned tiles.  usage: bash tools/profile_round.sh r01h
TAG=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python bench.py --tune-cache $O/${TAG}_tune.json > $O/${TAG}_bench_stage1.json 2> $O/${TAG}_bench_err.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune-cache $O/${TAG}_tune.json > $O/${TAG}_bench_stage1_under_rocprof.json 2>/dev/null
cp /tmp/ps/*/*kernel_stats.csv $O/${TAG}_bench_stage1_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --tune-cache $O/${TAG}_tune.json > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --tune-cache $O/${TAG}_tune.json > /dev/null 2>&1
python3 $R/tools/make_pmc_traffic.py /tmp/pf/*/*counter_collection.csv /tmp/pw/*/*counter_collection.csv $O/${TAG}_pmc_traffic.json
cd $R
python bench.py --stage 2 --no-cpu-baseline > $O/${TAG}_bench_stage2.json 2>/dev/null
python bench.py --task infer > $O/${TAG}_bench_infer.json 2>/dev/null
