"""stand-alone A/B of conv tile candidates on one layer shape, cold and hot: python tools/ab_tile.py B H Cin Cout k tiles...
(GPU box).  Cold = 8 rotating operand sets (beyond the caches), hot = the same operands every launch."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd  # noqa: F401
from disyolo_amd import lib as L

B, H, Cin, Cout, k = (int(v) for v in sys.argv[1:6])
tiles = [int(t, 0) for t in sys.argv[6:]]
dev = torch.device("cuda:0")
bf = torch.bfloat16
NSET = 8
xs = [torch.randn(B, H, H, Cin, device=dev).to(bf) for _ in range(NSET)]
ws = [(torch.randn(Cout, k * k * Cin, device=dev) * 0.02).to(bf) for _ in range(NSET)]
ys = [torch.empty(B, H, H, Cout, dtype=bf, device=dev) for _ in range(NSET)]
sc = torch.ones(Cout, device=dev)
sh = torch.zeros(Cout, device=dev)
flops = 2.0 * B * H * H * Cout * Cin * k * k
for t in tiles:
    ds = [L.make_conv_desc(xs[i], ws[i], ys[i], k, 1, scale=sc, shift=sh, leaky=True, tile=t) for i in range(NSET)]
    got = L.conv2d_tile(ds[0])
    res = []
    for mode, n in (("cold", NSET), ("hot", 1)):
        for _ in range(3):
            for i in range(n):
                L.conv2d_fwd(ds[i])
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(4):
                for i in range(n):
                    L.conv2d_fwd(ds[i])
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (4 * n) * 1e3)
        res.append(best)
    print("tile %#6x (runs %s): cold %.1f us = %.0f TFLOP/s | hot %.1f us = %.0f TFLOP/s"
          % (t, got, res[0], flops / res[0] / 1e6, res[1], flops / res[1] / 1e6), flush=True)
