"""Where a patch-kernel launch spends its time (GPU box; the library built with -DHALO_PROBE, see tools/README.md):
DISYOLO_LIB=dis-yolo_amd/libdisyolo_probe.so python tools/probe_halo.py
Per wave: s_memtime at kernel entry / first DMAs issued / main loop done / end, s_memrealtime at entry."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import disyolo_amd
from disyolo_amd import lib as L

dev = torch.device("cuda:0")
bf = torch.bfloat16
CASES = ((8, 18, 512, 1024, 24, 8), (8, 36, 256, 512, 25, 8), (8, 36, 512, 256, 24, 8), (8, 36, 256, 512, 16, 8), (8, 18, 512, 1024, 18, 8), (8, 72, 128, 256, 16, 8), (8, 18, 512, 1024, 16, 8),
         # the 192x128 GEMM tile (8 waves of 48x64, two blocks per CU): training and inference batch
         (8, 72, 128, 256, 12, 8), (32, 72, 128, 256, 12, 8), (32, 36, 256, 512, 12, 8), (32, 18, 512, 1024, 12, 8))
for (B, H, Cin, Cout, tile, nw) in CASES:
    x = torch.randn(B, H, H, Cin, device=dev).to(bf)
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(bf)
    y = torch.empty(B, H, H, Cout, dtype=bf, device=dev)
    sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    d = L.make_conv_desc(x, w, y, 3, 1, scale=sc, shift=sh, leaky=True, tile=tile)
    tid, bm, bn, bk, st = L.conv2d_tile(d)
    rows = L.conv2d_stats_rows(d)
    nblk = rows * (-(-Cout // bn))
    probe = torch.zeros(nblk * nw * 8, dtype=torch.int64, device=dev)
    flush = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
    for rep in range(3):
        flush.fill_(1)
        d.flags |= 0x200000
        d.stats = probe.data_ptr()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); L.conv2d_fwd(d); e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3
    t = probe.cpu().numpy().reshape(nblk, nw, 8)
    t0, t1, t2, t3, rt = (t[:, :, k].astype(np.float64) for k in range(5))
    ent = rt - rt.min()                                   # 100 MHz ticks
    print("B=%d %d^2 %d->%d tile %d (%dx%d): %d blocks, kernel %.1f us (events, cold)" % (B, H, Cin, Cout, tid, bm, bn, nblk, us))
    print("   block entry after the first: median %.2f us, max %.2f us" % (np.median(ent.min(1)) / 100, ent.min(1).max() / 100))
    print("   per wave, ticks at 2.4 GHz: setup %.0f | main loop %.0f | epilogue %.0f | total %.0f (= %.2f us)"
          % (np.median(t1 - t0), np.median(t2 - t1), np.median(t3 - t2), np.median(t3 - t0), np.median(t3 - t0) / 2400))
    print("   last block ends %.2f us after the first entry (realtime of entry + its own duration)" % ((ent + (t3 - t0) / 24).max() / 100))
    order = np.sort(ent.min(1))
    print("   block entries (us after the first), deciles: " + " ".join("%.1f" % (order[int(q * (len(order) - 1))] / 100) for q in np.linspace(0, 1, 11)))
    K = 9 * Cin
    print("   MFMA time of the block's main loop at the pipe's rate: %.0f ticks per SIMD (%d px x %d ch x K %d)" % (bm * bn * K / (16 * 16 * 32) / 4 * 16.3, bm, bn, K))
