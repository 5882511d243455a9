"""Tile sweep over the data-gradient conv shapes (3x3 stride-1 layers with Cin/Cout swapped):
python tools/bench_dgrad_shapes.py [tiles]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
tiles = [int(t, 0) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 3, 0x203, 10, 12, 0x108, 1, 2, 6]
dev = torch.device("cuda:0")
B = 8
SHAPES = [(144, 128, 64, 3), (72, 256, 128, 3), (36, 512, 256, 3), (18, 1024, 512, 3),
          (72, 128, 256, 1), (36, 256, 512, 1), (18, 512, 1024, 1), (36, 256, 768, 1), (72, 128, 384, 1)]
print("%-22s" % "shape (H,Cin,Cout,k)" + " ".join("%8s" % ("t%x" % t) for t in tiles))
for H, cin, cout, k in SHAPES:
    x0 = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, k * k * cin, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(B, H, H, cout, dtype=torch.bfloat16, device=dev)
    row = []
    for t in tiles:
        d = L.make_conv_desc(x0, w, y, k, 1, tile=t)
        try:
            for _ in range(3):
                L.conv2d_fwd(d)
        except Exception:
            row.append("       -")
            continue
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            L.conv2d_fwd(d)
        e.record()
        torch.cuda.synchronize()
        dt = s.elapsed_time(e) / 20 * 1e-3
        row.append("%8.1f" % (2.0 * B * H * H * cout * cin * k * k / dt / 1e12))
    print("%-22s" % str((H, cin, cout, k)) + " ".join(row))
