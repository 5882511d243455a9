"""the detection filter alone (decode + per-class greedy NMS + merge) on random logits: an untrained network's candidate counts
(python tools/bench_detect.py [B] [S])"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import disyolo_amd
from disyolo_amd import lib as L, config as cfg
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 576
for shift in (-4.0, 0.0, 1.5, 4.0):
    g = torch.Generator().manual_seed(0)
    logits = []
    for gs in (S // 8, S // 16, S // 32):
        y = torch.randn(B, gs, gs, 3, 8, generator=g)
        y[..., 4] += shift
        logits.append(y.reshape(B, gs, gs, 24).contiguous().to(dev))
    win = torch.tensor([[0.0, 0.0, 1.0, 1.0]] * B, device=dev)
    det = torch.zeros(B, cfg.MAX_DETECTION, 6, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    ws = L.Workspace(dev)
    anchors = np.asarray(cfg.ANCHORS, np.float32).reshape(-1)
    def run():
        L.detect(logits[0], logits[1], logits[2], B, S, 3, anchors, win, cfg.OBJ_THRESHOLD, cfg.IOU_THRESHOLD, cfg.MAX_DETECTION, det, cnt, ws)
    run(); torch.cuda.synchronize()
    NC = 3 * sum(gs * gs for gs in (S // 8, S // 16, S // 32))
    sc = ws.buf[B * NC * 16:B * NC * 20].view(torch.float32)
    npass = int((sc > cfg.OBJ_THRESHOLD).sum()) / B
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        run()
    e.record(); torch.cuda.synchronize()
    print("B=%d %d^2 conf shift %+.1f: %.0f of %d candidates per image over the threshold, detect %.1f us" % (B, S, shift, npass, NC, s.elapsed_time(e) * 1e3 / 20), flush=True)
