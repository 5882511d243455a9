"""which trainable layers run batch norm inside their conv launch, and what a layer's forward costs either way
(GPU box): python tools/bn_inkernel_probe.py [stage]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd  # noqa: F401
from disyolo_amd import lib as L
from disyolo_amd import config as cfg
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
net = YOLONet(training=True, device=dev, image_size=576, batch_size=8, stage=stage, seed=0)
net.set_batch(synthetic_batch(8, 576, seed=1))
cache = os.path.join(ROOT, "profiles", "tune_train_B8_576_stage%d.json" % stage)
net.autotune(cache=cache)
net.train_step()
torch.cuda.synchronize()


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print("%-4s %-22s %-28s %5s %6s | %9s %9s %9s" % ("idx", "shape", "tile", "rows", "fused", "fused us", "3-launch", "conv only"))
tot = [0.0, 0.0]
for l in net.layers:
    if l.lock or l.kind == "lin":
        continue
    M = net.B * l.Ho * l.Wo
    tile = L.conv2d_tile(l.desc)
    x0 = net._input_of(l, l.src)
    x1 = net._input_of(l, l.src_up) if l.src_up is not None else None
    plain = L.make_conv_desc(x0, l.wp, l.raw, l.k, l.stride, x1=x1, stats=l.stats)

    def three():
        L.conv2d_fwd(plain)
        L.bn_finalize(l.stats, l.stats_rows, l.cout, M, l.gamma, l.beta, l.mm, l.mv, cfg.BN_DECAY, cfg.BN_EPSILON, l.scale, l.shift, l.mean, l.rstd)
        L.bn_act_fwd(l.raw, l.scale, l.shift, None, l.act, M, l.cout, cfg.ALPHA)
    t3 = timed(three)
    tc = timed(lambda: L.conv2d_fwd(plain))
    tf = timed(lambda: L.conv2d_fwd(l.desc)) if l.fused_fwd else float("nan")
    print("%-4d %-22s %-28s %5d %6s | %9.1f %9.1f %9.1f" % (l.idx, "%dx%d %d->%d k%d" % (l.Ho, l.Wo, l.cin, l.cout, l.k), str(tile), l.stats_rows,
                                                          l.fused_fwd, tf, t3, tc), flush=True)
    if l.fused_fwd:
        tot[0] += tf
        tot[1] += t3
print("fused layers: %.1f us fused, %.1f us as three launches" % tuple(tot))
for l in net.layers:
    if l.csync is not None:
        assert L.cluster_sync_error(l.csync, l.cout) == 0, l.idx
