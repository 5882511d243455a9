"""Greedy in-STEP tile sweep (GPU box, one process): starting from a committed table, for every conv shape of the training
step try each tile that covers it IN the real two-lane step (re-recorded per candidate) and keep a change only when the
whole step gets faster by more than the noise floor, confirmed by a second measurement against a fresh base.
usage: python tools/instep_sweep.py [--stage 1] [--table profiles/tune_train_B8_576_stage1.json] [--out gpurun_out/sweep.json]
       [--min-gain 0.003] [--only k3|k1|all]"""
import argparse, gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import disyolo_amd
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
from bench import synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=1)
ap.add_argument("--table", default=os.path.join(ROOT, "profiles", "tune_train_B8_576_stage1.json"))
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
ap.add_argument("--min-gain", type=float, default=0.0015)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--budget", type=float, default=1500.0, help="seconds")
ap.add_argument("--only", default="all")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=576)
ap.add_argument("--skip-forward", action="store_true")
ap.add_argument("--ab", default=None, help="'<key json>=<cand>,<cand>,...': interleaved rounds of these candidates only")
ap.add_argument("--compare", default=None, help="another table (or sweep output): interleaved rounds of --table and this one")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--start", type=int, default=0, help="resume at this key index (keys are ordered by size)")
ap.add_argument("--plain", action="store_true", help="stage 1: sweep inside the plain (overlapped-tail) step instead of the pipelined one")
args = ap.parse_args()
dev = torch.device("cuda:0")
B, S = args.batch, args.size
net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=args.stage, seed=0)
net.set_batch(synthetic_batch(B, S, seed=1234))
torch.manual_seed(1234)
tj = json.load(open(args.table))
if "table" in tj:                    # a previous sweep's output: continue from its table
    args.table = os.path.join(ROOT, "gpurun_out", "_sweep_in.json")
    json.dump(tj["table"], open(args.table, "w"))
picks = net.autotune(cache=args.table)
net.shuffle_seed = 1234
table = dict(picks)


def descs():
    for l in net.layers:
        for name in ("desc", "wgrad_desc"):
            d = getattr(l, name, None)
            if d is not None and hasattr(d, "ksize"):
                yield d
        for d in (getattr(l, "dgrad_descs", None) or []):
            if hasattr(d, "ksize"):
                yield d


def rebuild():
    net.sync_lanes()
    torch.cuda.synchronize()
    net._prog = net._prog_marks = None
    gc.collect()                 # (the dropped command list's events and side streams are released in its destructor)
    net.ws.frozen = False
    net.ws_aux.frozen = False
    L.TUNED.clear()
    L.TUNED.update({k: v for k, v in table.items() if v})
    net._apply_tiles()
    net._progs = None
    if args.stage == 1 and not args.plain:
        # round 6: the step bench.py and Solver run in stage 1 is the PIPELINED one (next batch's backbone on the third lane)
        net.build_program(pipeline_backbone=True)
        net.prime_pipeline()
    else:
        net.build_program(overlap_tail=True)


def measure(reps=3):
    for _ in range(6):
        net.train_step(None, want_loss=False)
    ts = []
    for _ in range(reps):
        net.sync_lanes(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            net.train_step(None, want_loss=False)
        net.sync_lanes(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / args.steps * 1e3)
    return min(ts)


if args.compare:
    tj = json.load(open(args.compare))
    other = {tuple(json.loads(k)): int(v) for k, v in tj.get("table", tj).items()}
    mine = dict(table)
    res = {"a": [], "b": []}
    for r in range(args.rounds):
        for name, t in (("a", mine), ("b", other)):
            table.clear(); table.update(t)
            rebuild()
            res[name].append(round(measure(3), 4))
    diff = {json.dumps(list(k)): (mine.get(k), other.get(k)) for k in set(mine) | set(other) if mine.get(k) != other.get(k)}
    print("differences:", diff)
    print("a = %s: %s median %.4f" % (args.table, res["a"], float(np.median(res["a"]))))
    print("b = %s: %s median %.4f" % (args.compare, res["b"], float(np.median(res["b"]))), flush=True)
    sys.exit(0)
if args.ab:
    kj, cs = args.ab.split("=")
    key = tuple(json.loads(kj))
    cands = [int(c, 0) for c in cs.split(",")]
    res = {c: [] for c in cands}
    for r in range(args.rounds):
        for c in cands:
            table[key] = c
            rebuild()
            res[c].append(round(measure(3), 4))
    for c in cands:
        print("%s tile %#x: %s  median %.4f" % (key, c, res[c], float(np.median(res[c]))), flush=True)
    sys.exit(0)
rebuild()
base = measure(5)
print("base %.4f ms/step, %d shapes" % (base, len(table)), flush=True)
resolved = {}
for d in descs():
    resolved.setdefault(L.conv_shape_key(d), d)
for k in table:                      # the data-gradient descriptors are built while recording: a stand-in with the key's geometry
    if k not in resolved:
        d = L.ConvDesc()
        (d.B, d.H, d.W, d.C0, d.C1, d.Ho, d.Wo, d.Cout, d.ksize, d.stride, d.in_div) = k
        d.pad_t = d.pad_l = (d.ksize - 1) // 2
        d.flags = 0
        resolved[k] = d
keys = [k for k in table if k in resolved]
if args.skip_forward:
    fwd = set(L.conv_shape_key(d) for d in descs())
    keys = [k for k in keys if k not in fwd]
if args.only == "k3":
    keys = [k for k in keys if k[8] == 3]
elif args.only == "k1":
    keys = [k for k in keys if k[8] == 1]
# biggest layers first (2*M*N*K)
keys.sort(key=lambda k: -(k[0] * k[5] * k[6] * k[7] * k[8] * k[8] * (k[3] + k[4])))
t_start = time.time()
log = []
keys = keys[args.start:]


def applicable(d, cand):
    keep = d.tile
    d.tile = cand
    tid = L.conv2d_tile(d)[0]
    d.tile = keep
    return (not cand) or tid == (cand & 0xff)


def timed(key, cand, reps=2):
    table[key] = cand
    try:
        rebuild()
        return measure(reps)
    except Exception as e:      # noqa
        print("  ", key, hex(cand), "failed:", str(e)[:100], flush=True)
        return float("inf")


def save():
    json.dump({"table": {json.dumps(list(k)): v for k, v in table.items()}, "log": log}, open(args.out, "w"))


for n, key in enumerate(keys):
    if time.time() - t_start > args.budget:
        print("budget reached at key index %d" % (args.start + n), flush=True)
        break
    cur = table[key]
    d = resolved[key]
    cands = [c for c in (0,) + tuple(L.TUNE_CANDIDATES) if c != cur and (c & 0xff) != 20 and applicable(d, c)]
    if not applicable(d, cur):
        cur = 0                      # a table entry that does not cover the shape IS the heuristic
        cands = [c for c in cands if c != 0]
    # pass 1: every candidate once; pass 2: the best three and the current pick in interleaved rounds
    first = {c: timed(key, c) for c in [cur] + cands}
    short = sorted(cands, key=first.get)[:3]
    rounds = {c: [] for c in [cur] + short}
    for r in range(args.rounds):
        for c in [cur] + short:
            rounds[c].append(round(timed(key, c), 4))
    med = {c: float(np.median(v)) for c, v in rounds.items()}
    best = min(med, key=med.get)
    line = {"key": list(key), "current": cur, "first": {hex(c): round(t, 4) for c, t in first.items()},
            "rounds": {hex(c): v for c, v in rounds.items()}}
    if best != cur and med[best] < med[cur] * (1 - args.min_gain):
        line["kept"] = best
        table[key] = best
    else:
        table[key] = cur
    log.append(line)
    print(json.dumps(line), flush=True)
    save()
rebuild()
print("final %.4f ms/step" % measure(5), flush=True)
save()
