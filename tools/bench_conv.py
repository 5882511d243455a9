"""Per-layer micro-benchmark of the implicit-GEMM conv (GPU box): every distinct layer shape
of the 576x576 network x candidate tile configs -> TFLOP/s, to tune the launcher heuristic."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
from disyolo_amd.net import build_topology

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 576
tiles = [int(t, 0) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3, 6, 7, 8]
dev = torch.device("cuda:0")
layers = build_topology(3, 3)
spatial = {0: S}
shapes = {}
for l in layers:
    H = spatial[l.src]
    Ho, _ = L.same_pads(H, l.k, l.stride)
    spatial[l.idx] = Ho
    if l.idx == 1:
        continue
    key = (H, l.cin, l.cout, l.k, l.stride, l.src_up is not None)
    shapes.setdefault(key, []).append(l.idx)

_flush = None
def timeit(d, n=20):
    """cold-cache timing when COLD=1: a 512 MiB write between launches evicts L2 and the Infinity Cache"""
    global _flush
    if os.environ.get("COLD") == "1":
        if _flush is None:
            _flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        L.conv2d_fwd(d)
        tot = 0.0
        for _ in range(5):
            _flush.fill_(1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); L.conv2d_fwd(d); e.record(); torch.cuda.synchronize()
            tot += s.elapsed_time(e)
        return tot / 5 * 1e-3
    return _timeit_hot(d, n)


def _timeit_hot(d, n=20):
    for _ in range(3):
        L.conv2d_fwd(d)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        L.conv2d_fwd(d)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

print("%-34s %-10s %8s | " % ("shape (H,Cin,Cout,k,s)", "layers", "GFLOP") + " ".join("%9s" % ("t%x" % t) for t in tiles) + " | auto")
tot = {t: 0.0 for t in tiles}
tot_best = 0.0
tot_auto = 0.0
for key, idxs in sorted(shapes.items(), key=lambda kv: -kv[0][0]):
    H, cin, cout, k, s, fused = key
    Ho, _ = L.same_pads(H, k, s)
    if fused:
        c0 = cin * 2 // 3
        x0 = torch.randn(B, H, H, c0, device=dev).to(torch.bfloat16)
        x1 = torch.randn(B, H // 2, H // 2, cin - c0, device=dev).to(torch.bfloat16)
    else:
        x0 = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
        x1 = None
    w = (torch.randn(cout, k * k * cin, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(B, Ho, Ho, cout, dtype=torch.bfloat16, device=dev)
    sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
    fl = 2.0 * B * Ho * Ho * cout * cin * k * k
    row = []
    best = 1e9
    for t in tiles:
        if (cout <= 32 and (t & 0xff) in (1, 3, 8)) or (cout <= 64 and (t & 0xff) in (1, 3, 8)) or (cout > 64 and (t & 0xff) in (4, 5)):
            row.append("        -"); continue
        d = L.make_conv_desc(x0, w, y, k, s, x1=x1, scale=sc, shift=sh, leaky=True, tile=t)
        dt = timeit(d)
        best = min(best, dt)
        tot[t] += dt * len(idxs)
        row.append("%9.1f" % (fl / dt / 1e12))
    d = L.make_conv_desc(x0, w, y, k, s, x1=x1, scale=sc, shift=sh, leaky=True, tile=0)
    dta = timeit(d)
    tid = L.conv2d_tile(d)[0]
    tot_auto += dta * len(idxs)
    tot_best += min(best, dta) * len(idxs)
    print("%-34s %-10s %8.2f | " % (str(key[:5]) + ("F" if fused else ""), ("x%d" % len(idxs)), fl / 1e9) + " ".join(row) +
          " | %6.1f (t%d)" % (fl / dta / 1e12, tid))
print("forward conv total (layers 2..82): auto %.3f ms, best-of-candidates %.3f ms" % (tot_auto * 1e3, tot_best * 1e3))
