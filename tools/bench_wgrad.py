"""Per-layer micro-benchmark of the weight-gradient kernel (GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
from disyolo_amd.net import build_topology

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 576
variants = [int(t, 0) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]   # opts: 1 = im2col kernel (DISYOLO_WGRAD_IM2COL), 0 = planner's pick
first = int(sys.argv[4]) if len(sys.argv) > 4 else 53
dev = torch.device("cuda:0")
layers = build_topology(3, 3)
spatial = {0: S}
shapes = {}
for l in layers:
    H = spatial[l.src]
    Ho, _ = L.same_pads(H, l.k, l.stride)
    spatial[l.idx] = Ho
    if l.idx < max(first, 2):
        continue
    shapes.setdefault((H, l.cin, l.cout, l.k, l.stride), []).append(l.idx)
ws = L.Workspace(dev)
ws.get(1 << 28)
tot = {v: 0.0 for v in variants}
print("%-30s %-6s %8s | " % ("shape", "n", "GFLOP") + " ".join("%8s" % ("t%x" % v) for v in variants) + "  (TFLOP/s, us)")
for key, idxs in sorted(shapes.items(), key=lambda kv: -kv[0][0]):
    H, cin, cout, k, s = key
    Ho, _ = L.same_pads(H, k, s)
    x0 = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    ld = cout if cout % 8 == 0 else 32
    dy = torch.randn(B, Ho, Ho, ld, device=dev).to(torch.bfloat16)
    dw = torch.empty(k, k, cin, cout, device=dev)
    yd = torch.empty(B, Ho, Ho, cout, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * B * Ho * Ho * cout * cin * k * k
    row = []
    for v in variants:
        d = L.make_conv_desc(x0, dy, yd, k, s)
        for _ in range(3):
            L.conv2d_wgrad(d, dy, ld, dw, ws, opts=v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.conv2d_wgrad(d, dy, ld, dw, ws, opts=v)
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / 20 * 1e-3
        tot[v] += dt * len(idxs)
        row.append("%8.1f %6.1f" % (fl / dt / 1e12, dt * 1e6))
    print("%-30s x%-5d %8.2f | " % (str(key), len(idxs), fl / 1e9) + " ".join(row))
print("total ms: " + "  ".join("t%x %.3f" % (v, tot[v] * 1e3) for v in variants))
