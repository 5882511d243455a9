"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/<tag>_pmc_traffic.json
python tools/make_pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, sys, json, collections
def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        name = name.split("(")[0]
        tot[name] += float(r["Counter_Value"]); n[name] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}
f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
import subprocess, os
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__)),
                                     stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    commit = os.environ.get("DISYOLO_COMMIT", "working tree")
out = {"measured_at": (sys.argv[4] if len(sys.argv) > 4 else commit),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-cpu-baseline --no-secondary "
                 "--no-box` (the bench's own default steps / warm-up / repeats, committed tile table); per-kernel mean over all its launches (KiB); "
                 "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section), WRITE_SIZE as is; "
                 "hbm_mb_per_launch_corrected = (2*fetch_kb + write_kb) * 1024 / 1e6",
       "kernels": {}}
for k in sorted(f):
    if k in w:
        out["kernels"][k] = {"launches_sampled": f[k][1], "fetch_kb_raw": round(f[k][0], 1), "write_kb": round(w[k][0], 1),
                             "hbm_mb_per_launch_corrected": round((2 * f[k][0] + w[k][0]) * 1024 / 1e6, 2)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(len(out["kernels"]), "kernels")
