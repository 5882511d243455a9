"""Canaries for the regression round 2 shipped (Weak #1): the stage-2 train step at the BASELINE size
went to NaN after one step once the conv tiles had been autotuned.  Cause: the weight-gradient launcher read the
descriptor's ``tile`` field -- which belongs to the forward conv and which the tuner fills per GEMM shape -- as its own
timing switches; a tuned code with bit 9 set ("alternative pipeline depth") made it skip the slab reduction, so that
layer's dW was never written again and Adam kept applying whatever the buffer held (the non-finite leftovers of the
tuning passes, which ran on wrong batch statistics).  Every test here runs at B=8, 576x576 with tiles tuned the way
``bench.py`` tunes them -- the configuration none of the round-2 tests covered.
"""
import os
import sys

import numpy as np
import pytest
import torch

from disyolo_amd import config as cfg
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def clean_tune_table():
    L.TUNED.clear()
    yield
    L.TUNED.clear()
    L.TUNER = None


@pytest.mark.parametrize("tile", [0x202, 0x20c, 0x10c, 0x403, 0x12, 3])
@pytest.mark.parametrize("shape", [(2, 36, 64, 128, 3, 1), (2, 36, 128, 64, 1, 1), (2, 36, 64, 128, 3, 2)])
def test_weight_gradient_ignores_the_forward_tile_code(dev, tile, shape):
    """any code a tuner may leave in desc.tile: the weight gradient is bit-identical to tile = 0"""
    B, H, cin, cout, k, s = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, H, cin, generator=g).to(torch.bfloat16).to(dev)
    Ho, _ = L.same_pads(H, k, s)
    dy = torch.randn(B, Ho, Ho, cout, generator=g).to(torch.bfloat16).to(dev)
    y = torch.empty_like(dy)
    w = torch.zeros(cout, k * k * cin, dtype=torch.bfloat16, device=dev)
    out = []
    for t in (0, tile):
        dw = torch.full((k, k, cin, cout), float("nan"), device=dev)
        L.conv2d_wgrad(L.make_conv_desc(x, w, y, k, s, tile=t), dy, cout, dw, L.Workspace(dev))
        torch.cuda.synchronize()
        out.append(dw)
    assert bool(torch.isfinite(out[0]).all())
    assert torch.equal(out[0], out[1])


def _tuned_net(dev, stage, cache=None):
    B, S = 8, 576
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=stage, seed=0)
    net.set_batch(synthetic_batch(B, S, seed=1234))
    picks = net.autotune(cache=cache)
    net.shuffle_seed = 1234
    return net, picks


@pytest.mark.parametrize("stage", [1, 2])
def test_tuned_recorded_training_at_config_size_stays_finite_and_descends(dev, stage):
    """autotune() + build_program() + 60 recorded steps at B=8 / 576x576: every loss finite, the last below the
    first, every tuning pass itself finite, and every trainable variable's gradient is rewritten by every step"""
    net, picks = _tuned_net(dev, stage)
    # the tuning passes ran the real network (right batch statistics for every candidate tile): finite gradients
    assert bool(torch.isfinite(net.grad_arena).all()), "the autotuner's passes left non-finite gradients"
    net.build_program()
    losses = []
    for i in range(60):
        if i in (0, 1, 59):
            net.grad_arena.zero_()
        losses.append(float(net.train_step(None).cpu()))
        if i in (0, 1, 59):
            torch.cuda.synchronize()
            for name, (o, c) in net.arena_slices.items():
                assert float(net.grad_arena[o:o + c].abs().max()) > 0, "step %d wrote no gradient for %s" % (i, name)
    assert np.all(np.isfinite(losses)), "non-finite loss at step %d (tiles %s)" % (
        int(np.argmin(np.isfinite(losses))), {k: hex(v) for k, v in picks.items() if v})
    assert losses[-1] < losses[0], (losses[0], losses[-1])
    assert bool(torch.isfinite(net.arena).all())


@pytest.mark.parametrize("cache", [None, "profiles/archive/r02c_tune_cache.json"])
def test_fused_bn_backward_sums_match_the_plain_reduction_at_config_size(dev, cache):
    """every layer whose batch-norm backward sums come out of the data-gradient conv's epilogue (stage 2: the 1x1
    layers in front of the residual 3x3 convs, 288^2/32 ch ... 18^2/512 ch), with tuned tiles: the plain column
    reduction on the same gradient gives the same dgamma / dbeta / dx"""
    net, _ = _tuned_net(dev, 2, cache=os.path.join(ROOT, cache) if cache else None)
    net.bn_fuse_check = []
    net.train_step(None)
    torch.cuda.synchronize()
    rows = net.bn_fuse_check
    net.bn_fuse_check = None
    assert len(rows) >= 10, "only %d layers took the fused path" % len(rows)
    for r in rows:
        assert r["finite"], r
        # f32 sums in a different order; dx is bf16 (an ulp flips where the correction terms differ in the last bit)
        assert r["dgamma"] < 2e-4 and r["dbeta"] < 2e-4 and r["dx"] < 2e-3, r


def test_every_tile_candidate_agrees_on_every_launch_of_the_stage2_step(dev):
    """all 17 tile codes the tuner may pick x all distinct conv launches of a B=8 / 576x576 stage-2 step (forward with
    batch-norm partial sums, data gradients, accumulating and transposed ones), on integer-valued operands where any
    summation order is exact: bit-identical outputs, equal partial sums"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import tile_check
    lines = []
    nbad, n = tile_check.run(2, 8, 576, out=lines.append)
    assert n >= 50
    assert nbad == 0, "\n".join(ln for ln in lines if ln.startswith("BAD"))


def test_bench_fails_on_a_non_finite_loss(dev):
    """bench.py's canary: with a NaN planted in one weight the run exits with code 3, prints an "error" and no
    throughput; the same command without it exits 0 with a finite loss_first / loss_last"""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "2", "--size", "64", "--steps", "2", "--warmup", "2",
           "--repeats", "2", "--no-secondary", "--no-cpu-baseline", "--no-kernel-events", "--autotune", "off"]
    ok = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    line = json.loads([ln for ln in ok.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert ok.returncode == 0 and line["value"] > 0 and "error" not in line
    assert np.isfinite(line["config"]["loss_first"]) and np.isfinite(line["config"]["loss_last"])
    bad = subprocess.run(cmd + ["--poison"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    line = json.loads([ln for ln in bad.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert bad.returncode == 3 and line["value"] is None and "non-finite" in line["error"]
