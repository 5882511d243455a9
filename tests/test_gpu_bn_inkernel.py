"""Training-mode batch norm INSIDE the conv launch (DISYOLO_CONV_BN_FUSED / DISYOLO_CONV_BN_BWD_FUSED, csrc/conv_common.h
"cluster exchange") against the separate launches it replaces -- the same library, the same inputs:

  forward : conv (raw + statistics rows) -> disyolo_bn_finalize -> disyolo_bn_act_fwd
            (tf.nn.moments + batch_normalization + the moving-average assigns + leaky_relu, yolo/yolo3_net_pos.py:90-107)
  backward: data-gradient conv -> disyolo_bn_act_bwd of the layer it feeds (TF autodiff of the same lines)

The raw conv output must be bit-identical; scale / shift / mean / rstd / moving statistics agree to f32 rounding (the f64 sum
of the statistics rows runs in another order); the activation is then the same bf16 values except where a coefficient moved
by one ulp.  Every case runs MANY launches back to back on changing data -- the counters of the in-launch exchange re-arm
themselves -- with a second stream hammering HBM beside it (the hand-off must hold under uneven load, with warm caches),
and the error word of the bounded waits must stay zero."""
import numpy as np
import pytest
import torch

from disyolo_amd import lib as L

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
F32 = torch.float32

# B, H, Cin (C0, C1), Cout, k, tile -- the trainable layers of the 576^2 B = 8 step at 18^2 / 36^2 / 72^2, their table tiles
FWD_CASES = [
    (8, 18, (1024, 0), 512, 1, 3),        # conv53 / 55 / 57: 64x128 GEMM tiles, 41 rows, 164 blocks
    (8, 18, (1024, 0), 512, 1, 2),        # ... 128x64 tiles, 21 rows, 168 blocks
    (8, 18, (512, 0), 1024, 3, 18),       # conv54 / 56: patch kernel, 32 channels per block, 8 rows
    (8, 36, (256, 0), 512, 3, 16),        # conv62 / 64: patch kernel, 64 channels per block, 32 rows
    (8, 36, (512, 0), 256, 1, 10),        # conv63 / 65: 96x128 tiles, 108 rows
    (8, 36, (512, 256), 256, 1, 0x20c),   # conv61: fused upsample + concat, 192x128 tiles
    (8, 18, (512, 0), 256, 1, 0),         # conv60, the launcher's own pick (64x64, 164 blocks)
    (2, 18, (64, 0), 64, 3, 17),          # 4-wave patch kernel, a grid far smaller than the device
    (3, 20, (32, 0), 48, 1, 0),           # ragged: 1200 pixels, 48 channels (a partly empty channel tile)
]


def _mk(B, H, cin, cout, k, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    c0, c1 = cin
    x0 = (torch.randn(B, H, H, c0, device="cuda", generator=g) * 1.5).to(BF)
    x1 = (torch.randn(B, H // 2, H // 2, c1, device="cuda", generator=g)).to(BF) if c1 else None
    w = (torch.randn(cout, k * k * (c0 + c1), device="cuda", generator=g) * (1.0 / np.sqrt(k * k * (c0 + c1)))).to(BF)
    gamma = torch.rand(cout, device="cuda", generator=g) + 0.5
    beta = torch.randn(cout, device="cuda", generator=g) * 0.3
    return x0, x1, w, gamma, beta


class _Hammer:
    """HBM traffic on another stream while the launches under test run"""

    def __init__(self):
        self.s = torch.cuda.Stream()
        self.a = torch.empty(192 << 20, dtype=torch.uint8, device="cuda")
        self.b = torch.empty_like(self.a)

    def kick(self, n=2):
        with torch.cuda.stream(self.s):
            for _ in range(n):
                self.b.copy_(self.a)


@pytest.mark.parametrize("B,H,cin,cout,k,tile", FWD_CASES)
def test_forward_fused_matches_the_three_launches(dev, B, H, cin, cout, k, tile):
    M = B * H * H
    x0, x1, w, gamma, beta = _mk(B, H, cin, cout, k, 11)
    raw_a = torch.zeros(B, H, H, cout, dtype=BF, device=dev)
    raw_b = torch.zeros_like(raw_a)
    act_a = torch.zeros_like(raw_a)
    act_b = torch.full_like(raw_a, 7.0)
    d0 = L.make_conv_desc(x0, w, raw_a, k, 1, x1=x1, tile=tile)
    rows = L.conv2d_stats_rows(d0)
    st_a = torch.zeros(rows, cout, 2, dtype=F32, device=dev)
    st_b = torch.zeros_like(st_a)
    da = L.make_conv_desc(x0, w, raw_a, k, 1, x1=x1, stats=st_a, tile=tile)
    assert L.conv2d_bn_fused_ok(da), "this case is meant to run the fused epilogue"
    outs_a = [torch.zeros(cout, dtype=F32, device=dev) for _ in range(4)]      # scale shift mean rstd
    outs_b = [torch.full((cout,), 3.0, dtype=F32, device=dev) for _ in range(4)]
    mm_a, mv_a = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    mm_b, mv_b = mm_a.clone(), mv_a.clone()
    sync = L.cluster_sync_buffer(cout, dev)
    db = L.make_conv_desc(x0, w, raw_b, k, 1, x1=x1, stats=st_b, tile=tile,
                          bn_fused=dict(y_act=act_b, gamma=gamma, beta=beta, mm=mm_b, mv=mv_b, scale=outs_b[0], shift=outs_b[1],
                                        mean=outs_b[2], rstd=outs_b[3], decay=0.997, eps=1e-5, sync=sync))
    assert L.conv2d_tile(da)[0] == L.conv2d_tile(db)[0]
    ham = _Hammer()
    g = torch.Generator(device="cuda").manual_seed(5)
    for it in range(24):
        if it:      # new data every launch: a stale statistics row of the previous launch would show
            x0.copy_((torch.randn(x0.shape, device=dev, generator=g) * (1.0 + 0.1 * it)).to(BF))
        if it % 3 == 0:
            ham.kick()
        L.conv2d_fwd(da)
        L.bn_finalize(st_a, rows, cout, M, gamma, beta, mm_a, mv_a, 0.997, 1e-5, *outs_a)
        L.bn_act_fwd(raw_a, outs_a[0], outs_a[1], None, act_a, M, cout, 0.1)
        L.conv2d_fwd(db)
        torch.cuda.synchronize()
        assert torch.equal(raw_a.view(torch.int16), raw_b.view(torch.int16)), "raw conv output differs (launch %d)" % it
        assert torch.equal(st_a, st_b), "statistics rows differ (launch %d)" % it
        for name, a, b in zip(("scale", "shift", "mean", "rstd"), outs_a, outs_b):
            torch.testing.assert_close(b, a, rtol=2e-6, atol=1e-7, msg=lambda m: "%s (launch %d): %s" % (name, it, m))
        torch.testing.assert_close(mm_b, mm_a, rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(mv_b, mv_a, rtol=2e-6, atol=1e-7)
        # the activation: identical wherever the coefficients are (the same bf16 input, the same f32 arithmetic)
        same_c = (outs_a[0] == outs_b[0]) & (outs_a[1] == outs_b[1])
        assert float(same_c.float().mean()) > 0.95
        ea = act_a.view(torch.int16)[..., same_c]
        eb = act_b.view(torch.int16)[..., same_c]
        assert torch.equal(ea, eb), "activation differs on channels with identical coefficients (launch %d)" % it
        torch.testing.assert_close(act_b.float(), act_a.float(), rtol=1e-2, atol=1e-3)
    assert L.cluster_sync_error(sync, cout) == 0
    assert int(sync.abs().sum()) == 0, "the counters must be back at zero after every launch"


def test_fused_ok_wants_one_block_per_cu(dev):
    """a grid of several small blocks per CU is refused even though it would be resident on an idle device: a partly placed
    grid of that kind sits on every CU and can wedge against another queue's kernel (profiles/r06_bn_inkernel.txt)"""
    x = torch.zeros(8, 18, 18, 512, dtype=BF, device=dev)
    w = torch.zeros(1024, 512, dtype=BF, device=dev)
    y = torch.zeros(8, 18, 18, 1024, dtype=BF, device=dev)
    assert not L.conv2d_bn_fused_ok(L.make_conv_desc(x, w, y, 1, 1, tile=6))       # 41 x 16 = 656 blocks of 64x64
    assert L.conv2d_bn_fused_ok(L.make_conv_desc(x, w, y, 1, 1, tile=10))          # 27 x 8 = 216 blocks of 96x128


def test_fused_ok_refuses_what_cannot_be_resident_or_has_no_epilogue(dev):
    # 288^2 at B = 8: 2,592+ blocks of a 64-wide tile -- many rounds of the device
    x = torch.zeros(8, 288, 288, 32, dtype=BF, device=dev)
    w = torch.zeros(64, 9 * 32, dtype=BF, device=dev)
    y = torch.zeros(8, 288, 288, 64, dtype=BF, device=dev)
    assert not L.conv2d_bn_fused_ok(L.make_conv_desc(x, w, y, 3, 1, tile=2))
    # the flat-frame kernel and the streaming 1x1 kernel have no such epilogue
    x = torch.zeros(8, 72, 72, 256, dtype=BF, device=dev)
    w = torch.zeros(128, 9 * 256, dtype=BF, device=dev)
    y = torch.zeros(8, 72, 72, 128, dtype=BF, device=dev)
    d = L.make_conv_desc(x, w, y, 3, 1, tile=25)
    if L.conv2d_tile(d)[0] == 25:
        assert not L.conv2d_bn_fused_ok(d)
    # f32 outputs (the heads) never
    y32 = torch.zeros(8, 72, 72, 128, dtype=F32, device=dev)
    assert not L.conv2d_bn_fused_ok(L.make_conv_desc(x, w, y32, 3, 1, out_f32=True))


def test_a_descriptor_with_the_flag_on_a_kernel_without_the_epilogue_is_an_error(dev):
    x = torch.zeros(8, 72, 72, 256, dtype=BF, device=dev)
    w = torch.zeros(128, 9 * 256, dtype=BF, device=dev)
    y = torch.zeros(8, 72, 72, 128, dtype=BF, device=dev)
    d0 = L.make_conv_desc(x, w, y, 3, 1, tile=25)
    if L.conv2d_tile(d0)[0] != 25:
        pytest.skip("the flat-frame kernel does not cover this shape here")
    st = torch.zeros(L.conv2d_stats_rows(d0), 128, 2, dtype=F32, device=dev)
    v = [torch.zeros(128, device=dev) for _ in range(6)]
    d = L.make_conv_desc(x, w, y, 3, 1, tile=25, stats=st,
                         bn_fused=dict(y_act=torch.zeros_like(y), gamma=v[0], beta=v[1], mm=None, mv=None, scale=v[2], shift=v[3],
                                       mean=v[4], rstd=v[5], decay=0.997, eps=1e-5, sync=L.cluster_sync_buffer(128, dev)))
    with pytest.raises(L.DisyoloError):
        L.conv2d_fwd(d)


# data-gradient convs of the step whose target is a batch-normalised layer: B, H, C of dy (the conv's input), C of the target,
# k of the layer whose data gradient this is, tile
BWD_CASES = [
    (8, 18, 512, 1024, 1, 10),       # dgrad of conv55 / 57 (1x1 1024 -> 512) -> batch-norm backward of conv54 / 56: 96x128, 216 blocks
    (8, 18, 512, 1024, 1, 1),        # ... 128x128, 168 blocks
    (8, 18, 1024, 512, 3, 19),       # dgrad of conv54 / 56 / 58 (3x3) -> conv53 / 55 / 57: 16-channel patch tiles
    (8, 18, 1024, 512, 3, 18),
    (8, 36, 512, 256, 3, 16),        # dgrad of conv62 / 64 / 66 -> conv61 / 63 / 65
    (8, 36, 256, 512, 1, 12),        # dgrad of conv63 / 65 -> conv62 / 64: 192x128, 216 blocks
    (8, 36, 256, 512, 1, 0x20c),
    (2, 18, 64, 96, 1, 3),           # a partly empty channel tile (64x128)
    (2, 18, 64, 96, 1, 2),           # (128x64)
    (2, 18, 64, 96, 1, 6),           # (64x64)
]


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("B,H,cdy,ctgt,k,tile", BWD_CASES)
def test_backward_fused_matches_dgrad_plus_bn_act_bwd(dev, B, H, cdy, ctgt, k, tile, accumulate):
    """dy [B,H,H,cdy] --(flipped-weight conv)--> gradient wrt the target's activation [B,H,H,ctgt] --> the target's
    batch-norm backward.  ``accumulate``: an earlier contribution already sits in the gradient buffer (residual)."""
    M = B * H * H
    g = torch.Generator(device="cuda").manual_seed(3)
    dy = (torch.randn(B, H, H, cdy, device=dev, generator=g) * 0.5).to(BF)
    w = (torch.randn(ctgt, k * k * cdy, device=dev, generator=g) * (1.0 / np.sqrt(k * k * cdy))).to(BF)
    raw = (torch.randn(B, H, H, ctgt, device=dev, generator=g) * 2.0 + 0.3).to(BF)
    prev = (torch.randn(B, H, H, ctgt, device=dev, generator=g) * 0.7).to(BF) if accumulate else None
    gamma = torch.rand(ctgt, device=dev, generator=g) + 0.5
    beta = torch.randn(ctgt, device=dev, generator=g) * 0.3
    # the target's forward statistics (any self-consistent set does)
    mean = raw.float().mean(dim=(0, 1, 2))
    var = raw.float().var(dim=(0, 1, 2), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma * rstd
    shift = beta - mean * scale
    grad_a = torch.zeros(B, H, H, ctgt, dtype=BF, device=dev)
    dx_a = torch.zeros_like(grad_a)
    dx_b = torch.full_like(grad_a, 5.0)
    dga, dba = torch.zeros(ctgt, device=dev), torch.zeros(ctgt, device=dev)
    dgb, dbb = torch.full((ctgt,), 9.0, device=dev), torch.full((ctgt,), 9.0, device=dev)
    ws = L.Workspace(dev)
    ws.get(int(L.load().disyolo_bn_act_bwd_workspace(M, ctgt)))
    da = L.make_conv_desc(dy, w, grad_a, k, 1, residual=prev, tile=tile)
    rows = L.conv2d_stats_rows(da)
    part = torch.zeros(rows * ctgt * 2, dtype=F32, device=dev)
    sync = L.cluster_sync_buffer(ctgt, dev)
    db = L.make_conv_desc(dy, w, dx_b, k, 1, residual=prev, tile=tile,
                          bn_bwd=(raw, scale, shift, mean, rstd, part, 0.1), bn_bwd_fused=dict(dgamma=dgb, dbeta=dbb, sync=sync))
    assert L.conv2d_bn_fused_ok(db), "this case is meant to run the fused epilogue"
    ham = _Hammer()
    for it in range(16):
        if it:
            dy.copy_((torch.randn(dy.shape, device=dev, generator=g) * (0.5 + 0.05 * it)).to(BF))
        if it % 3 == 0:
            ham.kick()
        L.conv2d_fwd(da)
        L.bn_act_bwd(grad_a, raw, scale, shift, mean, rstd, dx_a, dga, dba, M, ctgt, ws, 0.1)
        L.conv2d_fwd(db)
        torch.cuda.synchronize()
        # the sums are taken over the same bf16 gradient values in another order (f32 partials per block): the per-channel
        # results agree to f32 summation accuracy, dx to a bf16 ulp
        nrm = lambda t: float(t.double().norm())
        assert nrm(dgb - dga) <= 2e-5 * nrm(dga) + 1e-6, "dgamma (launch %d)" % it
        assert nrm(dbb - dba) <= 2e-5 * nrm(dba) + 1e-6, "dbeta (launch %d)" % it
        a, b = dx_a.float(), dx_b.float()
        assert torch.isfinite(b).all()
        differ = (a != b)
        assert float(differ.float().mean()) < 2e-2, "dx differs in %.3f %% of the elements" % (100 * float(differ.float().mean()))
        torch.testing.assert_close(b, a, rtol=1.6e-2, atol=1e-4 * float(a.abs().max()))
    assert L.cluster_sync_error(sync, ctgt) == 0
    assert int(sync.abs().sum()) == 0


def test_training_step_with_and_without_inkernel_batch_norm_agree(dev):
    """the recorded step at the bench shape's head sizes (B = 2, 288^2: 9^2 / 18^2 / 36^2 maps), with the in-launch batch norm
    (forward AND backward form) against the separate launches: three steps from a COMMON state 30 steps into training (variables,
    both Adam moments, step count) -- from the random initialisation Adam's first steps follow the sign of gradients that are
    mostly rounding noise, and two summation orders go two ways (tests/test_gpu_trajectory.py).  Losses and the three-step update
    agree to the reordering of f32 / f64 sums."""
    from disyolo_amd.net import YOLONet
    from disyolo_amd.synth import synthetic_batch
    batch = synthetic_batch(2, 288, seed=5)

    def make(on):
        net = YOLONet(training=True, device=dev, image_size=288, batch_size=2, stage=1, seed=0)
        net.bn_inkernel = on
        net.bn_inkernel_bwd = on
        net._apply_tiles()
        return net
    a = make(True)
    for _ in range(30):
        a.train_step(batch, det_thresh=0.3)
    torch.cuda.synchronize()
    b = make(False)
    b.load_state_dict({k: v.clone() for k, v in a.state_dict().items()})
    b.adam_m.copy_(a.adam_m)
    b.adam_v.copy_(a.adam_v)
    b.step_dev.copy_(a.step_dev)
    b.refresh_weights()
    w0 = a.arena.clone()
    assert torch.equal(w0, b.arena)
    la = [float(a.train_step(batch, det_thresh=0.3).cpu()) for _ in range(3)]
    lb = [float(b.train_step(batch, det_thresh=0.3).cpu()) for _ in range(3)]
    torch.cuda.synchronize()
    nf, nb = sum(1 for l in a.layers if l.fused_fwd), sum(1 for l in a.layers if l.fused_bwd)
    assert nf >= 8 and nb >= 6, "expected most head layers to run their batch norm in the conv launch (%d fwd, %d bwd)" % (nf, nb)
    assert not any(l.fused_fwd or l.fused_bwd for l in b.layers)
    for l in a.layers:
        for buf in (l.csync, l.csync_bwd):
            if buf is not None:
                assert L.cluster_sync_error(buf, l.cout) == 0
    np.testing.assert_allclose(la, lb, rtol=2e-3)
    ua, ub = (a.arena - w0).double(), (b.arena - w0).double()
    assert float((ua @ ub) / (ua.norm() * ub.norm())) > 0.995
    assert float((ua - ub).norm() / ub.norm()) < 0.1
    for n, p in a.params.items():
        if "moving" in n:
            torch.testing.assert_close(p, b.params[n], rtol=1e-3, atol=1e-4)
