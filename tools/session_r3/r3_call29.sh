export DISYOLO_LIB=$GRAFT_REPO_ROOT/tools/bin/libdisyolo_probe.so
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for shape in "36 512 256 1 1 6" "72 256 128 1 1 0x202"; do
  for pr in 0 0x100000 0x94000 0x14000 0x80000; do
    rm -rf /tmp/pp
    PROBE=$pr rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/tools/one_conv.py $shape 8 30 > /dev/null 2>&1
    python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "conv_igemm" in r["Name"]:
        print("$shape probe $pr: avg %.2f us min %.2f (calls %s)" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, r["Calls"]))
PY
  done
done
