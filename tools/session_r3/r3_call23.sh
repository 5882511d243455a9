export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for t in force20 force202; do
  rm -rf /tmp/f_$t
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/f_$t -- python3 $R/bench.py --steps 20 --warmup 4 --repeats 1 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache $R/tools/tune_$t.json > /tmp/f_$t.json 2>/dev/null
  python3 - <<PY
import csv,re,json,glob
f=glob.glob("/tmp/f_$t/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
print("$t", json.load(open("/tmp/f_$t.json"))["ms_per_step"], "ms/step under rocprof")
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::|void |HIP_vector_type<[^>]*>|\(.*$","",r['Name'])
    if "conv_stream" in n or "conv_igemm_kernel<128, 64, 2, 2, 32, 3, 3, 1>" in n:
        print("   ", n, "calls/step", int(r['Calls'])/24, "avg us", float(r['AverageNs'])/1e3, "min", float(r['MinNs'])/1e3, "max", float(r['MaxNs'])/1e3)
PY
done
cd $R
for t in force20 force202 force20 force202; do
python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache tools/tune_$t.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['ms_per_step'])"
done
