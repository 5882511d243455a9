#!/usr/bin/env python3
"""Every conv launch of one training step (forward, data gradients, accumulating data gradients) x every tile
candidate of the autotuner, on integer-valued operands where any summation order is exact: all candidates
must produce the bit-identical output (and the same batch-norm partial sums).  Prints the outliers.

    python tools/tile_check.py --stage 2 [--batch 8 --size 576]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import disyolo_amd  # noqa: F401
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch


class Collector:
    """stands in for the tuner: remembers every descriptor conv2d_fwd is called with"""

    def __init__(self):
        self.descs = {}
        self.stats_rows = {}

    def launch(self, d):
        key = L.conv_shape_key(d) + (d.flags, bool(d.residual), bool(d.scale), bool(d.shift), d.pad_t, d.pad_l)
        if key not in self.descs:
            self.descs[key] = L.ConvDesc.from_buffer_copy(d)
        L._check(L.load().disyolo_conv2d_fwd(C.byref(d), L._stream()), "conv2d_fwd")


def ints(shape, lo, hi, dtype, dev, g):
    return torch.randint(lo, hi + 1, shape, generator=g, device=dev).to(dtype)


def check_desc(d0, cands, dev, g):
    B, H, W, C0, C1, Ho, Wo, Cout, k = d0.B, d0.H, d0.W, d0.C0, d0.C1, d0.Ho, d0.Wo, d0.Cout, d0.ksize
    K = k * k * (C0 + C1)
    x0 = ints((B, H, W, C0), -2, 2, torch.bfloat16, dev, g)
    x1 = ints((B, H // 2, W // 2, C1), -2, 2, torch.bfloat16, dev, g) if C1 else None
    w = ints((Cout, K), -1, 1, torch.bfloat16, dev, g)
    res = ints((B, Ho, Wo, Cout), -3, 3, torch.bfloat16, dev, g) if d0.residual else None
    scale = ints((Cout,), 1, 2, torch.float32, dev, g) if d0.scale else None
    shift = ints((Cout,), -2, 2, torch.float32, dev, g) if d0.shift else None
    f32 = bool(d0.flags & L.CONV_OUT_F32)
    want_stats = bool(d0.flags & L.CONV_STATS)
    results = {}
    for cand in cands:
        y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        stats = None
        d = L.make_conv_desc(x0, w, y, k, d0.stride, x1=x1, scale=scale, shift=shift, residual=res, leaky=bool(d0.flags & L.CONV_LEAKY),
                             out_f32=f32, alpha=d0.alpha, tile=cand, in_div=d0.in_div, pads=(d0.pad_t, d0.pad_l), out_hw=(Ho, Wo))
        if cand == 0:
            d.tile = 0
        if want_stats:
            rows = L.conv2d_stats_rows(d)
            stats = torch.full((rows, Cout, 2), float("nan"), dtype=torch.float32, device=dev)
            d = L.make_conv_desc(x0, w, y, k, d0.stride, x1=x1, scale=scale, shift=shift, residual=res, leaky=bool(d0.flags & L.CONV_LEAKY),
                                 out_f32=f32, alpha=d0.alpha, tile=cand, in_div=d0.in_div, pads=(d0.pad_t, d0.pad_l), out_hw=(Ho, Wo),
                                 stats=stats)
            if cand == 0:
                d.tile = 0
        rc = L.load().disyolo_conv2d_fwd(C.byref(d), L._stream())
        torch.cuda.synchronize()
        if rc != 0:
            results[cand] = ("error %d %s" % (rc, L.load().disyolo_last_error().decode()), None, None)
            continue
        tid = L.conv2d_tile(d)
        yy = y.view(torch.int32 if f32 else torch.int16)
        h = int(yy.to(torch.int64).sum().item()) ^ int((yy.to(torch.int64) * torch.arange(yy.numel(), device=dev).view(yy.shape) % 1000003).sum().item())
        nan = int(torch.isnan(y.float()).sum().item())
        st = stats.double().sum(0).cpu() if stats is not None else None
        results[cand] = (h, nan, st, tid)
    return results


def run(stage=2, batch=8, size=576, out=print):
    """returns (number of launches with a disagreeing candidate, number of distinct launches)"""
    dev = torch.device("cuda:0")
    net = YOLONet(training=True, device=dev, image_size=size, batch_size=batch, stage=stage, seed=0)
    net.set_batch(synthetic_batch(batch, size, seed=1234))
    col = Collector()
    L.TUNER = col
    net.compute_losses()
    net.backward()
    L.TUNER = None
    torch.cuda.synchronize()
    del net
    out("%d distinct conv launches" % len(col.descs))
    g = torch.Generator(device=dev).manual_seed(1)
    cands = (0,) + tuple(L.TUNE_CANDIDATES)
    nbad = 0
    for key, d0 in col.descs.items():
        res = check_desc(d0, cands, dev, g)
        groups = {}
        for cand, r in res.items():
            groups.setdefault(r[0], []).append(cand)
        major = max(groups.values(), key=len)
        ref = res[major[0]]
        bad = []
        for cand, r in res.items():
            why = None
            if r[0] != ref[0]:
                why = "output differs" if not isinstance(r[0], str) else r[0]
            elif r[1]:
                why = "%d NaN" % r[1]
            elif r[2] is not None and not torch.allclose(r[2], ref[2], rtol=1e-6, atol=1e-3):
                why = "stats differ (max rel %.3g)" % float(((r[2] - ref[2]).abs() / (ref[2].abs() + 1)).max())
            if why:
                bad.append((hex(cand), r[3] if len(r) > 3 else None, why))
        tag = "BAD" if bad else "ok "
        nbad += bool(bad)
        out("%s B%d H%d W%d C0=%d C1=%d -> Ho%d Wo%d Cout=%d k%d s%d in_div%d flags=%d res=%d sc=%d sh=%d pad=(%d,%d) nan_ref=%s %s"
            % ((tag,) + tuple(key) + (ref[1], bad if bad else "")))
    out("RESULT: %d of %d launches have a disagreeing candidate" % (nbad, len(col.descs)))
    return nbad, len(col.descs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=576)
    a = ap.parse_args()
    nbad, _ = run(a.stage, a.batch, a.size, out=lambda m: print(m, flush=True))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
