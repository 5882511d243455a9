#!/bin/bash
# kernel stats of the B=32 inference bench (GPU box).  usage: tools/infer_profile.sh <tag>
tag=$1
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ip_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ip_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --task infer --steps 10 --warmup 3 --repeats 2 --mode program > $O/${tag}_infer_under_rocprof.json 2>/dev/null
cp $(find /tmp/ip_$tag -name '*kernel_stats.csv' | head -1) $O/${tag}_infer_kernel_stats.csv
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("$O/${tag}_infer_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:22]:
    n=re.sub(r"\(anonymous namespace\)::|void |HIP_vector_type<[^>]*>|\(.*$","",r['Name'])
    print(f"{n[:60]:60s} calls={r['Calls']:>6} avg={float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.1f}%")
PY
