"""Per-lane timeline of ONE recorded step from a rocprofv3 --kernel-trace CSV:
python tools/timeline.py <kernel_trace.csv>   (uses the last complete step = between two optimizer-finish kernels)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "adam_fused_tail_kernel" in r["Kernel_Name"]]   # one per step (the sweeps are many)
i0, i1 = (adam[-3], adam[-2]) if len(adam) >= 3 else (adam[-2], adam[-1])
# a step = kernels after adam(i0)'s pack kernel .. adam(i1) + pack
step = rows[i0 + 1:i1 + 1]
t0 = min(r["s"] for r in step); t1 = max(r["e"] for r in step)
print("step wall %.1f us, %d kernels" % ((t1 - t0) / 1e3, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
def busy(rs):
    iv = sorted((r["s"], r["e"]) for r in rs); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for q, rs in byq.items():
    print("queue %s: %d kernels, busy %.1f us, span %.1f..%.1f" % (q, len(rs), busy(rs) / 1e3, (rs[0]["s"] - t0) / 1e3, (max(r["e"] for r in rs) - t0) / 1e3))
print("any-lane busy %.1f us" % (busy(step) / 1e3))
# phases by marker kernels on the main queue
def first(name):
    for r in step:
        if name in r["Kernel_Name"]: return (r["s"] - t0) / 1e3
    return None
for name in ("conv_first", "yolo_loss_kernel", "bn_bwd_finalize", "adam_fused_tail_kernel"):
    print("first %-20s at %8.1f us" % (name, first(name) or -1))
# per-queue per-kernel-class totals
for q, rs in byq.items():
    cls = collections.Counter()
    for r in rs:
        n = r["Kernel_Name"]
        k = "igemm" if "conv_igemm" in n else "halo" if "conv_halo" in n else "wgrad" if "conv_wgrad" in n else "slab" if "slab_reduce" in n else "bn" if ("bn_" in n or "colreduce" in n) else "other"
        cls[k] += (r["e"] - r["s"]) / 1e3
    print("queue", q, {k: round(v) for k, v in cls.items()})
# gaps on the main queue > 3 us
mainq = max(byq, key=lambda q: len(byq[q]))
rs = sorted(byq[mainq], key=lambda r: r["s"])
gaps = [(rs[i + 1]["s"] - rs[i]["e"], rs[i]["Kernel_Name"][:50], rs[i + 1]["Kernel_Name"][:50]) for i in range(len(rs) - 1)]
print("main queue idle total %.1f us; gaps > 4 us:" % (sum(max(g[0], 0) for g in gaps) / 1e3))
for g in sorted(gaps, reverse=True)[:12]:
    print("  %.1f us after %s -> %s" % (g[0] / 1e3, g[1], g[2]))
# backward window: per-queue busy
tb = [r for r in step if "bn_bwd_finalize" in r["Kernel_Name"]][0]["s"]
ta = [r for r in step if "adam_fused_tail_kernel" in r["Kernel_Name"]][0]["s"]
for q, rs in byq.items():
    w = [r for r in rs if r["s"] >= tb and r["e"] <= ta]
    f = [r for r in rs if r["e"] <= tb]
    if w: print("queue %s backward window: busy %.1f of %.1f us (%d kernels)" % (q, busy(w) / 1e3, (ta - tb) / 1e3, len(w)))
    if f: print("queue %s forward+loss window: busy %.1f of %.1f us (%d kernels)" % (q, busy(f) / 1e3, (tb - t0) / 1e3, len(f)))
# optional: dump the step's kernels in start order (argv[2] = output path)
if len(sys.argv) > 2:
    import re
    with open(sys.argv[2], "w") as f:
        for r in sorted(step, key=lambda r: r["s"]):
            n = re.sub(r"\(anonymous namespace\)::|void |HIP_vector_type<[^>]*>|\(.*$", "", r["Kernel_Name"])
            f.write("%9.1f %7.1f q%s %s grid=%s wg=%s\n" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["Queue_Id"], n[:70],
                                                      r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
