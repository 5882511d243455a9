"""In-STEP tile A/B (GPU box): the in-sequence tuner times candidates on one lane; kernels that hold a whole CU (144 KB of LDS)
behave differently beside the side lane's weight gradients.  For each (shape key -> candidate list) this replaces ONE entry of
the committed table, runs the real two-lane bench step and reports ms/step.
usage: python tools/instep_tune.py '<json: {"[8, 288, ...]": [516, 514], ...}>' [bench args]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
over = json.load(open(sys.argv[1][1:])) if sys.argv[1].startswith("@") else json.loads(sys.argv[1])
extra = sys.argv[2:]
base_p = os.environ.get("INSTEP_BASE", os.path.join(ROOT, "profiles", "tune_train_B8_576_stage1.json"))
base = json.load(open(base_p))
tmp = os.path.join(ROOT, "gpurun_out", "instep_table.json")

def run(table):
    json.dump(table, open(tmp, "w"))
    vals = []
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--repeats", "7", "--no-secondary",
                            "--no-box", "--no-cpu-baseline", "--no-kernel-events", "--tune-cache", tmp] + extra, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
        vals.append(d["ms_per_step"])
    return vals

print("base", run(dict(base)), flush=True)
for key, cands in over.items():
    assert key in base, key
    for c in cands:
        t = dict(base); t[key] = c
        print(key, "%d (%#x) instead of %d:" % (c, c, base[key]), run(t), flush=True)
print("base again", run(dict(base)), flush=True)
