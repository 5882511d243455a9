# one gpurun call: bench line + rocprofv3 kernel stats + PMC traffic for the SAME tuned tiles.  usage: bash tools/profile_round.sh r02c <label>
TAG=$1; LABEL=${2:-$1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
# tile cache first (so every pass runs the same tiles), then the PMC passes, then the bench line that reads the fresh traffic file
python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache $O/${TAG}_tune.json > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary --tune-cache $O/${TAG}_tune.json > $O/${TAG}_bench_stage1_under_rocprof.json 2>/dev/null
cp /tmp/ps/*/*kernel_stats.csv $O/${TAG}_bench_stage1_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache $O/${TAG}_tune.json > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache $O/${TAG}_tune.json > /dev/null 2>&1
python3 $R/tools/make_pmc_traffic.py /tmp/pf/*/*counter_collection.csv /tmp/pw/*/*counter_collection.csv $O/${TAG}_pmc_traffic.json "$LABEL"
cp $O/${TAG}_pmc_traffic.json $R/profiles/r02_pmc_traffic.json
cd $R
python bench.py --tune-cache $O/${TAG}_tune.json > $O/${TAG}_bench_stage1.json 2> $O/${TAG}_bench_err.txt
tail -c 3000 $O/${TAG}_bench_stage1.json
