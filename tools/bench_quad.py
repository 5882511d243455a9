"""stride-2 data gradients of the shallow layers: the quad kernel against the generic in_div = 2 (parity-class) path,
cold caches (GPU box): python tools/bench_quad.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
dev = torch.device("cuda:0")
bf = torch.bfloat16
flush = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / n * 1e3
for (B, H, C, Cdy) in ((8, 576, 32, 64), (8, 288, 64, 128)):
    dy = torch.randn(B, H // 2, H // 2, Cdy, device=dev).to(bf)
    w = torch.randn(3, 3, C, Cdy, device=dev) * 0.05
    wq = torch.zeros(4 * C, 9 * Cdy, dtype=bf, device=dev)
    L.pack_quad(w, wq)
    dx = torch.empty(B, H, H, C, dtype=bf, device=dev)
    tq = timeit(lambda: L.dgrad_s2_quad(dy, wq, dx))
    wdg = torch.empty(C, 9 * Cdy, dtype=bf, device=dev)
    L.pack_weights(w, None, wdg, 3, C, Cdy, Cdy)
    best = None
    for t in (4, 0x204, 2, 0x202, 6, 12, 3):
        d = L.make_conv_desc(dy, wdg, dx, 3, 1, in_div=2, pads=(2, 2), out_hw=(H, H), tile=t)
        tt = timeit(lambda: L.conv2d_fwd(d))
        best = (tt, t) if best is None or tt < best[0] else best
    mb = (B * (H // 2) ** 2 * Cdy + B * H * H * C) * 2 / 1e6
    print("dx %dx%dx%dx%d from %d channels: quad %.1f us (%.2f TB/s); generic parity-class path best %.1f us (tile 0x%x)"
          % (B, H, H, C, Cdy, tq, mb / tq, best[0], best[1]))
