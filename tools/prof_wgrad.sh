#!/bin/bash
# per-kernel durations of the weight-gradient kernels (run on the GPU box): tools/prof_wgrad.sh <tag> [env assignments...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o wg -- python3 tools/bench_wgrad.py 8 576 0 53 > $out.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
echo "== $tag $@"; tail -3 $out.log
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-90s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
