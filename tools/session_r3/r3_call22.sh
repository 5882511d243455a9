mkdir -p gpurun_out/r3
bash tools/onelane_stats.sh r3b_s1 --stage 1 > gpurun_out/r3/onelane_b_s1.txt 2>&1
cat gpurun_out/r3/onelane_b_s1.txt
python3 - <<'PY'
import csv,re
rows=list(csv.DictReader(open("gpurun_out/ol_r3b_s1_kernel_stats.csv")))
for r in rows[:40]:
    n=re.sub(r"\(anonymous namespace\)::|void |HIP_vector_type<[^>]*>|\(.*$","",r['Name'])
    print(f"{n[:64]:64s} calls/step={int(r['Calls'])/24:6.2f} avg={float(r['AverageNs'])/1e3:7.1f}us us/step={float(r['TotalDurationNs'])/24e3:7.1f}")
PY
