"""per-wave cycle stamps of the GEMM-tile kernel on the deep 1x1 layers (probe build: tools/build_probe.sh g_probe conv_igemm.hip "-DHALO_PROBE";
DISYOLO_LIB=dis-yolo_amd/libdisyolo_g_probe.so python tools/probe_1x1.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import disyolo_amd
from disyolo_amd import lib as L
dev = torch.device("cuda:0"); bf = torch.bfloat16
for (B, H, Cin, Cout, tile, nw) in ((8, 18, 1024, 512, 6, 4), (8, 36, 512, 256, 6, 4), (8, 72, 256, 128, 3, 4), (8, 18, 1024, 512, 0x206, 4)):
    x = torch.randn(B, H, H, Cin, device=dev).to(bf)
    w = (torch.randn(Cout, Cin, device=dev) * 0.02).to(bf)
    y = torch.empty(B, H, H, Cout, dtype=bf, device=dev)
    sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    d = L.make_conv_desc(x, w, y, 1, 1, scale=sc, shift=sh, leaky=True, tile=tile)
    tid, bm, bn, bk, st = L.conv2d_tile(d)
    nblk = (-(-B * H * H // bm)) * (-(-Cout // bn))
    probe = torch.zeros(nblk * nw * 8, dtype=torch.int64, device=dev)
    for rep in range(3):
        d.flags |= 0x200000
        d.stats = probe.data_ptr()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); L.conv2d_fwd(d); e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3
    t = probe.cpu().numpy().reshape(nblk, nw, 8)
    t0, t1, t2, t3, rt = (t[:, :, k].astype(np.float64) for k in range(5))
    ent = rt - rt.min()
    print("B=%d %d^2 %d->%d 1x1 tile %#x (%dx%d, BK %d, %d stages): %d blocks, kernel %.1f us (events, hot)" % (B, H, Cin, Cout, tile, bm, bn, bk, st, nblk, us))
    print("   per wave, cycles: setup %.0f | main loop %.0f | epilogue %.0f | total %.0f;  block entries up to %.2f us after the first; last end %.2f us"
          % (np.median(t1 - t0), np.median(t2 - t1), np.median(t3 - t2), np.median(t3 - t0), ent.min(1).max() / 100, (ent + (t3 - t0) / 22).max() / 100))
