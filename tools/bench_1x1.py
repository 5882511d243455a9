"""the deep 1x1 layers, kernel by kernel: the GEMM tiles the tables hold and their deep-pipeline forms, forward
(training statistics) and data-gradient (accumulate) forms, back-to-back launches over rotating buffers (python tools/bench_1x1.py [B])"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
dev = torch.device("cuda:0"); bf = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = 576
shapes = []
for H, a, b in ((S // 32, 1024, 512), (S // 16, 512, 256), (S // 8, 256, 128), (S // 16, 768, 256), (S // 8, 384, 128)):
    shapes.append((H, a, b, "fwd"))
    if a in (1024, 512, 256):
        shapes.append((H, b, a, "dgrad"))       # the 1x1 data gradient: Cout -> Cin with the transposed weights
tiles = (0, 6, 0x206, 3, 9, 0x209, 10, 11, 0x20b, 12)
NB = 6
for (H, Cin, Cout, form) in shapes:
    M = B * H * H
    xs = [torch.randn(B, H, H, Cin, device=dev).to(bf) for _ in range(NB)]
    w = (torch.randn(Cout, Cin, device=dev) * 0.02).to(bf)
    ys = [torch.zeros(B, H, H, Cout, dtype=bf, device=dev) for _ in range(NB)]
    line = []
    for tile in tiles:
        descs = []
        for i in range(NB):
            if form == "fwd":
                rows = L.conv2d_stats_rows(L.make_conv_desc(xs[i], w, ys[i], 1, 1, tile=tile))
                st = torch.zeros(rows, Cout, 2, device=dev)
                d = L.make_conv_desc(xs[i], w, ys[i], 1, 1, stats=st, tile=tile)
                d._keep = st
            else:
                d = L.make_conv_desc(xs[i], w, ys[i], 1, 1, residual=ys[i], tile=tile)
            descs.append(d)
        got = L.conv2d_tile(descs[0])[0]
        if tile and got != (tile & 0xff):
            line.append("%#x: n/a" % tile)
            continue
        for d in descs:
            L.conv2d_fwd(d)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 50
        s.record()
        for r in range(R):
            for d in descs:
                L.conv2d_fwd(d)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / (R * NB)
        line.append("%#x(%d): %.1f" % (tile, got, us))
    byt = (M * Cin + Cin * Cout + M * Cout * (2 if form == "dgrad" else 1)) * 2
    print("B=%d %d^2 %4d->%4d %-5s  %5.2f GFLOP %5.1f MB  us/launch: %s" % (B, H, Cin, Cout, form, 2e-9 * M * Cin * Cout, byt / 1e6, "  ".join(line)), flush=True)
