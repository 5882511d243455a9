"""the batch-norm passes of the big early layers, alone: forward normalise, backward column sums / finalize / apply, against the bytes
they move (python tools/bench_bn.py; DISYOLO_EXP_BN = 4 / 8 / 16 / 28 skips the column sums / finalize / apply / all three)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
dev = torch.device("cuda:0")
B = 8
ws = L.Workspace(dev)
for (S, C) in ((576, 32), (288, 64), (288, 32), (144, 128), (144, 64), (72, 256), (72, 128), (36, 512), (18, 1024)):
    rows = B * S * S
    NB = 3
    dy = [torch.randn(rows, C, device=dev).to(torch.bfloat16) for _ in range(NB)]
    x = [torch.randn(rows, C, device=dev).to(torch.bfloat16) for _ in range(NB)]
    dx = [torch.empty(rows, C, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    sc, sh, mu, rs = (torch.rand(C, device=dev) + 0.5 for _ in range(4))
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    def bwd(i):
        L.bn_act_bwd(dy[i], x[i], sc, sh, mu, rs, dx[i], dg, db, rows, C, ws)
    def fwd(i):
        L.bn_act_fwd(x[i], sc, sh, None, dx[i], rows, C)
    res = []
    for name, fn, nbytes in (("bn_act_bwd (3 kernels)", bwd, rows * C * 2 * 5), ("bn_act_fwd", fwd, rows * C * 2 * 2)):
        try:
            for i in range(NB):
                fn(i)
        except Exception as e:      # noqa
            res.append("%s: %s" % (name, str(e)[:40])); continue
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 10
        s.record()
        for r in range(R):
            for i in range(NB):
                fn(i)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / (R * NB)
        res.append("%s %.1f us = %.2f TB/s" % (name, us, nbytes / us / 1e6))
    print("B=8 %d^2 x %d (%.0f MB a tensor): %s" % (S, C, rows * C * 2 / 1e6, "; ".join(res)), flush=True)
