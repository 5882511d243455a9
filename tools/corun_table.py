"""Co-run table (VERDICT r5 task 5): the step's heaviest main-lane kernels beside its heaviest side-lane kernels -- A alone,
B alone, and both at once on two streams (the main lane's stream at normal priority, the side lane's at the lowest, as in the
recorded step) -- for B = 8, 576^2, stage 1 with the committed tile table (GPU box):

    python tools/corun_table.py > gpurun_out/r06_corun_table.txt

Per pair: time per launch of A and of B when each runs alone (back-to-back launches on its stream), and when both streams run
their loops at once (each loop sized for ~2 ms alone; the per-launch time is taken over the window in which BOTH loops are
still running).  "together / alone" = how much each kernel slows down beside the other; "work per us" = (tA_alone / tA_co +
tB_alone / tB_co): 1.0 = the two kernels simply share the chip (no gain from running them at once), 2.0 = they do not disturb
each other at all."""
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

dev = torch.device("cuda:0")
net = YOLONet(training=True, device=dev, image_size=576, batch_size=8, stage=1, seed=0)
net.bn_inkernel = False
net.set_batch(synthetic_batch(8, 576, seed=1))
net.autotune(cache=os.path.join(ROOT, "profiles", "tune_train_B8_576_stage1.json"))
net.train_step()
torch.cuda.synchronize()
by = net.by_idx
try:
    least, greatest = torch.cuda.Stream.priority_range()       # (lowest, highest): larger number = lower priority
except Exception:
    least, greatest = 0, 0
s_main = torch.cuda.Stream(priority=0)
s_side = torch.cuda.Stream(priority=least)    # the side lane of the recorded step has the lowest stream priority


def fwd(i):
    l = by[i]
    d = L.make_conv_desc(net._input_of(l, l.src), l.wp, l.raw, l.k, l.stride,
                         x1=net._input_of(l, l.src_up) if l.src_up is not None else None, stats=l.stats)
    return ("fwd conv%d %dx%d %d->%d k%d %s" % (i, l.Ho, l.Wo, l.cin, l.cout, l.k, L.conv2d_tile(d)[:3]), lambda: L.conv2d_fwd(d))


def backbone(i):
    l = by[i]
    return ("fwd conv%d (locked) %dx%d %d->%d k%d %s" % (i, l.Ho, l.Wo, l.cin, l.cout, l.k, L.conv2d_tile(l.desc)[:3]),
            lambda: L.conv2d_fwd(l.desc))


def wgrad(i):
    l = by[i]
    ws = L.Workspace(dev)
    ws.get(int(L.conv2d_wgrad_workspace(l.wgrad_desc)))
    return ("wgrad conv%d %dx%d %d->%d k%d" % (i, l.Ho, l.Wo, l.cin, l.cout, l.k), lambda: L.conv2d_wgrad(l.wgrad_desc, l.dx, l.cout, l.dw, ws))


def bn_apply(i):
    l = by[i]
    M = net.B * l.Ho * l.Wo
    ws = L.Workspace(dev)
    ws.get(int(L.load().disyolo_bn_act_bwd_workspace(M, l.cout)))
    return ("bn_act_bwd conv%d (%d MB tensor)" % (i, M * l.cout * 2 // 1000000),
            lambda: L.bn_act_bwd(l.grad, l.raw, l.scale, l.shift, l.mean, l.rstd, l.dx, l.dgamma, l.dbeta, M, l.cout, ws, cfg.ALPHA))


def adam():
    n = net.n_params
    return ("adam sweep, %.1f M variables" % (n / 1e6),
            lambda: L.adam_sweep(net.arena, net.grad_arena, net.adam_m, net.adam_v, n, net.n_decay, net.lr_dev, cfg.ADAM_BETA1, cfg.ADAM_BETA2,
                                 cfg.ADAM_EPSILON, 0.0, net.step_dev, 1.0, None))


MAIN = [backbone(12), backbone(29), fwd(54), fwd(62), fwd(70), fwd(53), fwd(71), fwd(81)]
SIDE = [wgrad(54), wgrad(62), wgrad(70), wgrad(57), wgrad(81), bn_apply(81), adam()]


def alone(fn, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def together(fa, fb, na, nb):
    """both loops at once; returns per-launch us of A and of B over the common window (the first min(na, nb-equivalent) launches)"""
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(na + 1)]
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(nb + 1)]
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    s_main.wait_event(start)
    s_side.wait_event(start)
    # interleave the enqueues so that neither stream runs ahead of the host
    ia = ib = 0
    with torch.cuda.stream(s_main):
        ea[0].record()
    with torch.cuda.stream(s_side):
        eb[0].record()
    while ia < na or ib < nb:
        if ia < na:
            with torch.cuda.stream(s_main):
                fa()
                ea[ia + 1].record()
            ia += 1
        if ib < nb:
            with torch.cuda.stream(s_side):
                fb()
                eb[ib + 1].record()
            ib += 1
    torch.cuda.synchronize()
    ta = [start.elapsed_time(e) for e in ea]
    tb = [start.elapsed_time(e) for e in eb]
    end = min(ta[-1], tb[-1])                      # the window in which both loops were running
    ka = max(1, sum(1 for t in ta[1:] if t <= end))
    kb = max(1, sum(1 for t in tb[1:] if t <= end))
    return (ta[ka] - ta[0]) / ka * 1e3, (tb[kb] - tb[0]) / kb * 1e3


print("%-58s %-44s | %8s %8s | %8s %8s | %6s %6s | %s" % ("main-lane kernel A", "side-lane kernel B", "A alone", "B alone", "A beside", "B beside",
                                                        "A x", "B x", "work per us"))
for na_, fa in MAIN:
    ta = alone(fa, s_main, 30)
    for nb_, fb in SIDE:
        tb = alone(fb, s_side, 30)
        na, nb = max(8, int(2000 / ta)), max(8, int(2000 / tb))
        best = None
        for _ in range(3):
            ca, cb = together(fa, fb, na, nb)
            if best is None or ca + cb < best[0] + best[1]:
                best = (ca, cb)
        ca, cb = best
        print("%-58s %-44s | %8.1f %8.1f | %8.1f %8.1f | %6.2f %6.2f | %.2f" % (na_[:58], nb_[:44], ta, tb, ca, cb, ca / ta, cb / tb, ta / ca + tb / cb), flush=True)
