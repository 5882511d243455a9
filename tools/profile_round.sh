# one gpurun call: bench line + rocprofv3 kernel stats + PMC traffic, all with the committed tile tables (bench.py --tune-cache auto).
# usage: bash tools/profile_round.sh r03a <label>     (outputs in gpurun_out/<tag>_*; copy what is to be kept into profiles/)
TAG=$1; LABEL=${2:-$1}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-box > $O/${TAG}_bench_stage1_under_rocprof.json 2>/dev/null
cp /tmp/ps/*/*kernel_stats.csv $O/${TAG}_bench_stage1_kernel_stats.csv
# PMC counters in their own passes (never together with a trace domain other than the kernel trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-box > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-box > /dev/null 2>&1
python3 $R/tools/make_pmc_traffic.py /tmp/pf/*/*counter_collection.csv /tmp/pw/*/*counter_collection.csv $O/${TAG}_pmc_traffic.json "$LABEL"
cp $O/${TAG}_pmc_traffic.json $R/profiles/r06_pmc_traffic.json     # (so that the bench line below already reads this round's traffic)
cd $R
python bench.py > $O/${TAG}_bench_stage1.json 2> $O/${TAG}_bench_err.txt
tail -c 4000 $O/${TAG}_bench_stage1.json
