#!/usr/bin/env python3
"""Per-layer forward times of the inference net (GPU box): python tools/infer_layers.py [B] [S] [dtype]
tuned tiles, eager launches, HIP events around every layer, median of 7 passes; beside each layer its algorithmic
FLOP and bf16 bytes (in + weights + out [+ residual]) and the time either would take at 2.5 PF / 8 TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import disyolo_amd  # noqa: F401
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 576
dev = torch.device("cuda:0")
net = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
b = synthetic_batch(B, S, seed=1234)
net._set_inputs(b["images"], b["clip_window"])
net.autotune()
net.use_side_lane = False
P = 7
ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in net.layers] for _ in range(P)]
plan = net._fusion_plan(False, 1, 82)          # fused launches: timed under the group's last layer, the others show 0
for p in range(P):
    for i, l in enumerate(net.layers):
        ev[p][i][0].record()
        if l.idx in plan:
            if plan[l.idx] is not None:
                plan[l.idx]()
        else:
            net._forward_layer(l, False)
        ev[p][i][1].record()
torch.cuda.synchronize()
tot = 0.0
tot_bound = 0.0
print("%3s %-22s %8s %8s %8s %8s %8s" % ("L", "shape", "us", "TF/s", "GB/s", "mfma_us", "hbm_us"))
groups = {}
for i, l in enumerate(net.layers):
    us = float(np.median([ev[p][i][0].elapsed_time(ev[p][i][1]) for p in range(P)])) * 1e3
    M = B * l.Ho * l.Wo
    K = l.k * l.k * l.cin
    fl = 2.0 * M * l.cout * K
    by = B * l.H * l.W * l.cin * (4 if l.idx == 1 else 2) + K * l.cout * 2 + M * l.cout * (4 if l.kind == "lin" else 2) * (2 if l.shortcut else 1)
    t_m, t_h = fl / 2.5e15 * 1e6, by / 8e12 * 1e6
    tot += us
    tot_bound += max(t_m, t_h)
    key = "%dx%d s%d" % (l.k, l.k, l.stride) + (" @%d" % l.Ho)
    groups.setdefault(key, [0, 0.0])
    groups[key][0] += 1
    groups[key][1] += us
    print("%3d %-22s %8.1f %8.1f %8.1f %8.1f %8.1f" % (l.idx, "%d->%d %dx%d/%d @%d" % (l.cin, l.cout, l.k, l.k, l.stride, l.Ho), us,
                                                     fl / us / 1e6, by / us / 1e3, t_m, t_h))
print("sum %.1f us (%.0f img/s network only); bound sum %.1f us" % (tot, B / tot * 1e6, tot_bound))
for k, (n, us) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print("  %-14s x%-3d %8.1f us" % (k, n, us))
