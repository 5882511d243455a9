"""Per-layer forward time of the backbone (conv1-52), bf16 kernels vs the fp8 path (GPU box).
usage: python tools/bench_fp8_layers.py [B] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 576
dev = torch.device("cuda:0")
b = synthetic_batch(B, S, seed=1)
nets = {}
for dt in ("bf16", "fp8"):
    n = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0, dtype=dt)
    n._set_inputs(b["images"], b["clip_window"])
    if dt == "bf16":
        n.autotune()
    else:
        n.calibrate_fp8()
    nets[dt] = n
tot = {"bf16": 0.0, "fp8": 0.0}
print("%-4s %-28s %9s %9s" % ("idx", "shape", "bf16 us", "fp8 us"))
for i in range(1, 53):
    row = []
    for dt in ("bf16", "fp8"):
        n = nets[dt]; l = n.by_idx[i]
        for _ in range(3):
            n._forward_layer(l, False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            n._forward_layer(l, False)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        tot[dt] += t; row.append(t)
    l = nets["bf16"].by_idx[i]
    print("%-4d %-28s %9.1f %9.1f" % (i, "%dx%d %d->%d k%d s%d" % (l.H, l.W, l.cin, l.cout, l.k, l.stride), row[0], row[1]))
print("total us: bf16 %.1f  fp8 %.1f" % (tot["bf16"], tot["fp8"]))
# the whole backbone as the product runs it (bf16: fused launches for conv1+2 / the 288^2 and 144^2 residual blocks; fp8: layer by layer)
for dt in ("bf16", "fp8"):
    n = nets[dt]
    for _ in range(3):
        n._forward_prefix(52, False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        n._forward_prefix(52, False)
    e1.record(); torch.cuda.synchronize()
    print("backbone conv1-52 as the product runs it, %s: %.1f us" % (dt, e0.elapsed_time(e1) / 20 * 1e3))
