"""GPU parity of the implicit-GEMM convolution family against the CPU oracle.

Tolerances: inputs/weights are rounded to bf16 once and shared by both sides, the
oracle computes in f64, the kernels accumulate in f32 on MFMA and (unless OUT_F32)
round the result to bf16.  bf16 has 8 significant bits, so |err| <= 2^-8 |y| + a small
absolute term for the f32 accumulation; f32 outputs are held to 1e-4 relative to the
largest output magnitude.
"""
import numpy as np
import pytest
import torch

import disyolo_oracle as O
from disyolo_amd import lib as L

pytestmark = pytest.mark.gpu


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float64)


def pack_ref(w_hwio):
    k, _, cin, cout = w_hwio.shape
    return w_hwio.permute(3, 0, 1, 2).reshape(cout, k * k * cin).contiguous()


def check(got, want, rel, abs_):
    got = got.double().cpu()
    want = want.detach().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all(), "non-finite values in kernel output"
    err = (got - want).abs()
    tol = rel * want.abs() + abs_
    excess = (err - tol).flatten()
    i = int(excess.argmax())
    assert excess[i] <= 0, "err %.4g > tol %.4g (%d of %d elems out of tolerance; max|want| %.4g)" % (
        float(err.flatten()[i]), float(tol.flatten()[i]), int((excess > 0).sum()), excess.numel(), float(want.abs().max()))


CASES = [
    # B, H, W, Cin, Cout, k, s, tile
    (2, 18, 18, 64, 128, 3, 1, 0),
    (2, 18, 18, 64, 128, 3, 1, 1),
    (2, 18, 18, 64, 128, 3, 1, 3),
    (1, 20, 20, 32, 64, 3, 2, 0),      # asymmetric SAME pads
    (1, 20, 20, 32, 64, 3, 2, 6),
    (2, 17, 19, 32, 64, 3, 2, 2),      # odd sizes: symmetric pads, ragged M
    (2, 12, 12, 128, 32, 1, 1, 0),
    (2, 12, 12, 96, 32, 1, 1, 4),
    (1, 40, 40, 64, 64, 3, 1, 7),
    (3, 9, 9, 256, 256, 3, 1, 0),
    (1, 16, 16, 1024, 512, 1, 1, 0),
    (2, 18, 18, 64, 128, 3, 1, 8),         # 256x128 tile, 8 waves
    (2, 18, 18, 64, 128, 3, 1, 0x101),     # BK forced to 32 on a BK=64-capable shape
    (2, 18, 18, 64, 128, 3, 1, 0x108),
    (4, 36, 36, 128, 256, 3, 1, 1),        # 36 K-steps: exercises the steady-state pipeline
    (4, 36, 36, 128, 256, 3, 1, 8),
    (1, 12, 12, 32, 32, 3, 1, 6),          # 9 K-steps of BK=32
    (2, 12, 12, 64, 16, 1, 1, 5),          # nk = 1 (shorter than the pipeline depth)
    (2, 12, 12, 128, 64, 1, 1, 2),         # nk = 2
    (2, 18, 18, 64, 128, 3, 1, 10),        # 96x128 tile (48x64 wave tiles), ragged last M tile
    (2, 18, 18, 96, 128, 3, 1, 0x10a),     # ... at BK=32 (A tile padded to the DMA slab)
    (4, 36, 36, 128, 256, 3, 1, 0x20b),    # ... deep pipeline variant
    (2, 12, 12, 128, 128, 1, 1, 11),
    (4, 36, 36, 128, 256, 3, 1, 12),       # 192x128 tile, 8 waves of 48x64
    (2, 18, 18, 96, 128, 3, 1, 0x10c),
    (2, 18, 18, 128, 256, 3, 1, 13),       # intra-block split-K (two K groups of 4 waves)
    (2, 18, 18, 128, 256, 3, 1, 0x20d),
    (4, 36, 36, 128, 256, 3, 1, 14),
    (2, 18, 18, 192, 128, 3, 1, 14),       # 27 K slices: falls back to the plain 96x128 tile
    (2, 18, 18, 192, 128, 1, 1, 13),       # 3 K slices: falls back
    (2, 18, 18, 384, 128, 3, 1, 13),       # 6 channel slices per tap, groups interleave within a tap
    (4, 36, 36, 128, 256, 3, 1, 15),       # 16 waves, all 160 KiB of LDS
    (1, 16, 16, 1024, 512, 1, 1, 14),      # 1x1, split-K
    (2, 12, 12, 96, 128, 3, 1, 14),        # falls back to the plain 96x128 tile (BK = 32)
    (2, 18, 18, 64, 128, 3, 1, 16),        # halo-reuse patch kernel: one 18x18 patch per image, 8 waves
    (1, 36, 36, 64, 64, 3, 1, 16),         # four patches per image
    (2, 18, 18, 64, 128, 3, 1, 17),        # 9x18 patches, 4 waves
    (1, 26, 26, 32, 96, 3, 1, 16),         # 13x26 patches (832^2 network), ragged channel tile, one K slice
    (2, 20, 24, 96, 64, 3, 1, 16),         # 20x12 patches, three K slices
    (1, 20, 20, 32, 64, 3, 2, 16),         # stride 2: not covered, falls back to the GEMM kernel
    (3, 9, 9, 256, 256, 3, 1, 16),         # 9x9 patch = whole image, 6 of 24 fragment slots used
    (2, 18, 18, 64, 128, 3, 1, 18),        # patch kernel, 32 output channels per block
    (1, 18, 18, 96, 72, 3, 1, 18),         # ... ragged channel tile (72 = 2 x 32 + 8), three K slices
    (2, 18, 18, 64, 128, 3, 1, 19),        # patch kernel, 16 output channels per block
    (1, 18, 18, 96, 72, 3, 1, 19),         # ... ragged channel tile (72 = 4 x 16 + 8)
    (2, 18, 18, 64, 128, 3, 1, 24),        # flat-frame patch kernel, 32x32x16 MFMA: 192 x 64 tiles, two K slices (two stages)
    (8, 18, 18, 128, 128, 3, 1, 24),       # ... four slices on three stages, 16 M tiles (the last one nearly empty)
    (3, 9, 11, 96, 64, 3, 1, 24),          # ... three slices, H != W, frame shorter than two tiles
    (1, 36, 36, 160, 192, 3, 1, 24),       # ... five slices (three-stage loop leaves mid-unroll), P = 37
    (2, 36, 36, 64, 128, 3, 1, 25),        # 384 x 64 tiles (3x2 fragments per wave)
    (1, 26, 52, 192, 64, 3, 1, 25),        # ... six slices, P = 53
    (1, 72, 72, 64, 64, 3, 1, 25),         # ... P = 73: nine DMA pieces per wave
    (1, 20, 20, 32, 64, 3, 2, 24),         # stride 2 / 32 input channels: not covered, falls back
    (2, 18, 18, 64, 72, 3, 1, 24),         # Cout not a multiple of 64: falls back
    (1, 16, 16, 1024, 512, 1, 1, 26),      # two-wave 32x64 tile on a deep 1x1 layer (16 K slices)
    (2, 18, 18, 512, 256, 1, 1, 0x21a),    # ... alternative pipeline depth, ragged last M tile (648 = 20 x 32 + 8)
    (2, 18, 18, 96, 72, 3, 1, 26),         # ... 3x3, BK = 32, ragged channel tile
    (1, 16, 16, 1024, 512, 1, 1, 27),      # 64x32 tile (two waves along M)
    (2, 17, 19, 32, 40, 3, 2, 27),         # ... stride 2, odd sizes, ragged M and N
    (2, 18, 18, 512, 256, 1, 1, 28),       # 32x128 tile (32x64 wave tiles)
    (2, 12, 12, 128, 64, 1, 1, 0x21c),     # ... nk = 2, fewer slices than stages; N smaller than the tile
    (4, 36, 36, 128, 256, 3, 1, 29),       # 384x128 tile, 8 waves of 96x64 (13.5 M tiles: ragged last one)
    (2, 18, 18, 96, 72, 3, 1, 29),         # ... BK = 32, ragged channel tile, M smaller than two tiles
    (1, 16, 16, 1024, 512, 1, 1, 29),      # ... 1x1, one M tile
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,tile", CASES)
def test_conv_fwd_matches_oracle(dev, B, H, W, Cin, Cout, k, s, tile):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin + Cout + k + s + tile)
    x = bf16r(torch.randn(B, H, W, Cin, generator=g))
    w = bf16r(torch.randn(k, k, Cin, Cout, generator=g) / (k * k * Cin) ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    want_raw = O.conv2d_same(x, w, s)
    want = O.leaky_relu(want_raw * scale.double() + shift.double(), 0.1)
    xd = x.to(torch.bfloat16).to(dev)
    wd = pack_ref(w).to(torch.bfloat16).to(dev)
    y = torch.empty(B, want.shape[1], want.shape[2], Cout, dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(xd, wd, y, k, s, scale=scale.to(dev), shift=shift.to(dev), leaky=True, tile=tile)
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    check(y, want, 2.0 ** -7, 1e-3)


def test_conv_identity_asymmetric(dev):
    """A = I check with an asymmetric operand (cdna guide s3): 1x1 conv with identity
    weights must reproduce the input exactly; catches row/col swaps in the MFMA maps."""
    B, H, W, Cch = 1, 16, 16, 128
    x = (torch.arange(B * H * W * Cch, dtype=torch.float32).reshape(B, H, W, Cch) % 251) - 125
    w = torch.eye(Cch).reshape(1, 1, Cch, Cch)
    xd = x.to(torch.bfloat16).to(dev)
    wd = pack_ref(w).to(torch.bfloat16).to(dev)
    y = torch.empty(B, H, W, Cch, dtype=torch.float32, device=dev)
    d = L.make_conv_desc(xd, wd, y, 1, 1, out_f32=True)
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x)


def test_conv_out_f32_bias_small_n(dev):
    """head-style conv: 1x1, bias, linear, f32 out, Cout = 24 and 9 (ragged N)."""
    for Cout, tile in ((24, 0), (9, 0), (9, 4)):
        g = torch.Generator().manual_seed(Cout)
        x = bf16r(torch.randn(2, 18, 18, 256, generator=g))
        w = bf16r(torch.randn(1, 1, 256, Cout, generator=g) / 16)
        bias = torch.randn(Cout, generator=g)
        want = O.conv2d_same(x, w, 1) + bias.double()
        y = torch.empty(2, 18, 18, Cout, dtype=torch.float32, device=dev)
        d = L.make_conv_desc(x.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev), y, 1, 1,
                             shift=bias.to(dev), out_f32=True, tile=tile)
        L.conv2d_fwd(d)
        torch.cuda.synchronize()
        check(y, want, 1e-5, 1e-4 * float(want.abs().max()))


@pytest.mark.parametrize("tile", [16, 17, 18, 19])
def test_conv_halo_residual_f32_and_pads(dev, tile):
    """patch kernel epilogue variants: residual add, f32 output with bias, and the data-gradient
    use (explicit pads, accumulate into an existing gradient through the residual pointer)"""
    g = torch.Generator().manual_seed(70 + tile)
    x = bf16r(torch.randn(2, 18, 36, 64, generator=g))
    w = bf16r(torch.randn(3, 3, 64, 72, generator=g) / 24)
    res = bf16r(torch.randn(2, 18, 36, 72, generator=g))
    xd, wd = x.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev)
    y = torch.empty(2, 18, 36, 72, dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(xd, wd, y, 3, 1, residual=res.to(torch.bfloat16).to(dev), leaky=True, tile=tile)
    assert L.conv2d_tile(d)[0] == tile
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    check(y, O.leaky_relu(O.conv2d_same(x, w, 1), 0.1) + res, 2.0 ** -7, 1e-3)
    bias = torch.randn(72, generator=g)
    yf = torch.empty(2, 18, 36, 72, dtype=torch.float32, device=dev)
    d = L.make_conv_desc(xd, wd, yf, 3, 1, shift=bias.to(dev), out_f32=True, tile=tile)
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    want = O.conv2d_same(x, w, 1) + bias.double()
    check(yf, want, 1e-5, 1e-4 * float(want.abs().max()))
    d = L.make_conv_desc(xd, wd, y, 3, 1, pads=(1, 1), out_hw=(18, 36), residual=y, tile=tile)   # y += conv
    before = y.float().cpu().double()
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    check(y, before + O.conv2d_same(x, w, 1), 2.0 ** -7, 2e-3)


@pytest.mark.parametrize("tile", [24, 25])
def test_conv_flat_residual_accumulate_and_integer_identity(dev, tile):
    """flat-frame patch kernels: residual add, the data-gradient use (y += conv through the residual pointer), and
    -- on integer-valued operands, where every summation order is exact -- the bit-identical result of a GEMM tile"""
    g = torch.Generator().manual_seed(170 + tile)
    B, H, W, Cin, Cout = 3, 18, 36, 128, 128
    x = bf16r(torch.randn(B, H, W, Cin, generator=g))
    w = bf16r(torch.randn(3, 3, Cin, Cout, generator=g) / 34)
    res = bf16r(torch.randn(B, H, W, Cout, generator=g))
    xd, wd = x.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev)
    y = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(xd, wd, y, 3, 1, residual=res.to(torch.bfloat16).to(dev), leaky=True, tile=tile)
    assert L.conv2d_tile(d)[0] == tile
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    check(y, O.leaky_relu(O.conv2d_same(x, w, 1), 0.1) + res, 2.0 ** -7, 1e-3)
    d = L.make_conv_desc(xd, wd, y, 3, 1, pads=(1, 1), out_hw=(H, W), residual=y, tile=tile)   # y += conv
    before = y.float().cpu().double()
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    check(y, before + O.conv2d_same(x, w, 1), 2.0 ** -7, 2e-3)
    xi = torch.randint(-3, 4, (B, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    wi = torch.randint(-2, 3, (Cout, 9 * Cin), generator=g).to(torch.bfloat16).to(dev)
    outs = []
    for t in (3, tile):
        yi = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
        L.conv2d_fwd(L.make_conv_desc(xi, wi, yi, 3, 1, tile=t))
        torch.cuda.synchronize()
        outs.append(yi.clone())
    assert float(outs[0].float().abs().max()) > 50 and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("B,H,W,Cout", [(2, 36, 36, 64), (3, 72, 48, 64), (1, 144, 144, 128), (2, 18, 90, 40)])
def test_conv_stream_kernel_32_input_channels(dev, B, H, W, Cout):
    """tile 20, the persistent streaming patch kernel (3x3 stride 1, 32 input channels): folded BN + leaky + residual,
    the training-mode form (raw output + per-patch statistics rows), an accumulating data gradient, ragged channel
    tiles (Cout 40), more patches than persistent blocks' first round and fewer -- against the f64 conv"""
    g = torch.Generator().manual_seed(B * 100 + H + Cout)
    x = bf16r(torch.randn(B, H, W, 32, generator=g))
    w = bf16r(torch.randn(3, 3, 32, Cout, generator=g) / 17)
    res = bf16r(torch.randn(B, H, W, Cout, generator=g))
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.3
    xd, wd = x.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev)
    y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(xd, wd, y, 3, 1, scale=scale.to(dev), shift=shift.to(dev), residual=res.to(torch.bfloat16).to(dev),
                         leaky=True, tile=20)
    assert L.conv2d_tile(d)[0] == 20
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    conv = O.conv2d_same(x, w, 1)
    check(y, O.leaky_relu(conv * scale.double() + shift.double(), 0.1) + res, 2.0 ** -7, 2e-3)
    # training-mode batch norm: raw output + statistics partials, one row per patch
    y2 = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
    rows = L.conv2d_stats_rows(L.make_conv_desc(xd, wd, y2, 3, 1, tile=20))
    stats = torch.full((rows, Cout, 2), float("nan"), dtype=torch.float32, device=dev)
    L.conv2d_fwd(L.make_conv_desc(xd, wd, y2, 3, 1, stats=stats, tile=20))
    torch.cuda.synchronize()
    check(y2, conv, 2.0 ** -7, 1e-3)
    flat = conv.reshape(-1, Cout)
    got = stats.double().sum(0).cpu()
    assert torch.allclose(got[:, 0], flat.sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(got[:, 1], (flat * flat).sum(0), rtol=1e-4, atol=1e-2)
    # the data-gradient use: explicit pads, accumulate into an existing gradient through the residual pointer
    before = y2.float().cpu().double()
    L.conv2d_fwd(L.make_conv_desc(xd, wd, y2, 3, 1, pads=(1, 1), out_hw=(H, W), residual=y2, tile=20))
    torch.cuda.synchronize()
    check(y2, before + conv, 2.0 ** -7, 3e-3)
    # bit-identical to the per-patch kernel on integer-valued operands (any summation order is exact)
    xi = torch.randint(-2, 3, (B, H, W, 32), generator=g).to(torch.bfloat16).to(dev)
    wi = torch.randint(-1, 2, (Cout, 288), generator=g).to(torch.bfloat16).to(dev)
    outs = []
    for tile in (20, 2):
        yo = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
        L.conv2d_fwd(L.make_conv_desc(xi, wi, yo, 3, 1, residual=res.to(torch.bfloat16).to(dev), leaky=True, tile=tile))
        torch.cuda.synchronize()
        outs.append(yo)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


@pytest.mark.parametrize("B,H,W,C0,C1,Cout", [(2, 36, 36, 64, 0, 32), (1, 72, 48, 128, 64, 64), (3, 24, 40, 64, 32, 32),
                                              (2, 50, 30, 64, 0, 9), (1, 144, 144, 128, 0, 64), (2, 18, 18, 32, 0, 64),
                                              (1, 30, 22, 128, 0, 40)])
def test_conv1x1_stream_kernel(dev, B, H, W, C0, C1, Cout):
    """tile 21, the wave-autonomous streaming 1x1 kernel: folded BN + leaky (+ residual), fused upsample + concat, the
    linear f32 form with a bias and 9 channels (conv82), training-mode statistics (one row per block), an accumulating
    data gradient, ragged pixel counts and channel tiles -- against the f64 conv"""
    g = torch.Generator().manual_seed(B * 100 + H + Cout + C1)
    x0 = bf16r(torch.randn(B, H, W, C0, generator=g))
    x1 = bf16r(torch.randn(B, H // 2, W // 2, C1, generator=g)) if C1 else None
    w = bf16r(torch.randn(1, 1, C0 + C1, Cout, generator=g) / (C0 + C1) ** 0.5)
    xin = torch.cat([x0, O.upsample2(x1)], -1) if C1 else x0
    conv = O.conv2d_same(xin, w, 1)
    x0d = x0.to(torch.bfloat16).to(dev)
    x1d = x1.to(torch.bfloat16).to(dev) if C1 else None
    wd = pack_ref(w).to(torch.bfloat16).to(dev)
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.3
    if Cout % 8 == 0:
        res = bf16r(torch.randn(B, H, W, Cout, generator=g))
        y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
        d = L.make_conv_desc(x0d, wd, y, 1, 1, x1=x1d, scale=scale.to(dev), shift=shift.to(dev), residual=res.to(torch.bfloat16).to(dev),
                             leaky=True, tile=21)
        assert L.conv2d_tile(d)[0] == 21
        L.conv2d_fwd(d)
        torch.cuda.synchronize()
        check(y, O.leaky_relu(conv * scale.double() + shift.double(), 0.1) + res, 2.0 ** -7, 2e-3)
        # training-mode batch norm: raw output + statistics partials
        y2 = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
        rows = L.conv2d_stats_rows(L.make_conv_desc(x0d, wd, y2, 1, 1, x1=x1d, tile=21))
        stats = torch.full((rows, Cout, 2), float("nan"), dtype=torch.float32, device=dev)
        L.conv2d_fwd(L.make_conv_desc(x0d, wd, y2, 1, 1, x1=x1d, stats=stats, tile=21))
        torch.cuda.synchronize()
        check(y2, conv, 2.0 ** -7, 1e-3)
        flat = conv.reshape(-1, Cout)
        got = stats.double().sum(0).cpu()
        assert torch.allclose(got[:, 0], flat.sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(got[:, 1], (flat * flat).sum(0), rtol=1e-4, atol=1e-2)
        if not C1:
            before = y2.float().cpu().double()
            L.conv2d_fwd(L.make_conv_desc(x0d, wd, y2, 1, 1, residual=y2, tile=21))       # y += conv: the data-gradient use
            torch.cuda.synchronize()
            check(y2, before + conv, 2.0 ** -7, 3e-3)
    # linear f32 output with a bias (conv59 / 67 / 75 / 82)
    yf = torch.full((B, H, W, Cout), float("nan"), dtype=torch.float32, device=dev)
    L.conv2d_fwd(L.make_conv_desc(x0d, wd, yf, 1, 1, x1=x1d, shift=shift.to(dev), out_f32=True, tile=21))
    torch.cuda.synchronize()
    want = conv + shift.double()
    check(yf, want, 1e-5, 1e-4 * float(want.abs().max()))
    # bf16 output with a ragged channel count goes element by element
    yb = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
    L.conv2d_fwd(L.make_conv_desc(x0d, wd, yb, 1, 1, x1=x1d, scale=scale.to(dev), shift=shift.to(dev), leaky=True, tile=21))
    torch.cuda.synchronize()
    check(yb, O.leaky_relu(conv * scale.double() + shift.double(), 0.1), 2.0 ** -7, 2e-3)


def test_conv_residual_and_fused_concat(dev):
    g = torch.Generator().manual_seed(7)
    # residual (res_conv_bn, yolo/yolo3_net_pos.py:148-151): add AFTER the activation
    x = bf16r(torch.randn(2, 18, 18, 64, generator=g))
    w = bf16r(torch.randn(3, 3, 64, 128, generator=g) / 24)
    res = bf16r(torch.randn(2, 18, 18, 128, generator=g))
    want = O.leaky_relu(O.conv2d_same(x, w, 1), 0.1) + res
    y = torch.empty(2, 18, 18, 128, dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(x.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev), y, 3, 1,
                         residual=res.to(torch.bfloat16).to(dev), leaky=True)
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    check(y, want, 2.0 ** -7, 1e-3)
    # fused nearest-upsample + concat [skip, up] feeding a 1x1 conv (:290-293)
    skip = bf16r(torch.randn(2, 12, 12, 64, generator=g))
    low = bf16r(torch.randn(2, 6, 6, 32, generator=g))
    w = bf16r(torch.randn(1, 1, 96, 32, generator=g) / 10)
    cat = torch.cat([skip, O.upsample2(low)], dim=-1)
    want = O.conv2d_same(cat, w, 1)
    y = torch.empty(2, 12, 12, 32, dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(skip.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev), y, 1, 1,
                         x1=low.to(torch.bfloat16).to(dev))
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    check(y, want, 2.0 ** -7, 1e-3)


@pytest.mark.parametrize("tile", [0, 10, 12, 13, 14, 16, 17, 24, 25])
def test_conv_stats_and_bn_finalize(dev, tile):
    """training BN: stats epilogue + finalize == tf.nn.moments (population variance) and the
    moving-average update of yolo/yolo3_net_pos.py:90-98."""
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = 2, 18, 18, 64, 128
    x = bf16r(torch.randn(B, H, W, Cin, generator=g))
    w = bf16r(torch.randn(3, 3, Cin, Cout, generator=g) / 24)
    raw = O.conv2d_same(x, w, 1)
    y = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(x.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev), y, 3, 1, tile=tile)
    rows = L.conv2d_stats_rows(d)
    stats = torch.zeros(rows, Cout, 2, dtype=torch.float32, device=dev)
    d = L.make_conv_desc(x.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev), y, 3, 1, stats=stats,
                         tile=tile)
    L.conv2d_fwd(d)
    gamma = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    beta = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    mm = torch.zeros(Cout, device=dev)
    mv = torch.ones(Cout, device=dev)
    scale, shift, mean, rstd = (torch.empty(Cout, device=dev) for _ in range(4))
    L.bn_finalize(stats, rows, Cout, B * H * W, gamma, beta, mm, mv, 0.997, 1e-5, scale, shift, mean, rstd)
    act = torch.empty_like(y)
    L.bn_act_fwd(y, scale, shift, None, act, B * H * W, Cout)
    torch.cuda.synchronize()
    m = raw.mean(dim=(0, 1, 2))
    v = ((raw - m) ** 2).mean(dim=(0, 1, 2))
    check(mean, m, 1e-4, 1e-5)
    check(rstd, 1 / torch.sqrt(v + 1e-5), 1e-4, 1e-5)
    check(mm, 0.003 * m, 1e-4, 1e-6)
    check(mv, 0.997 + 0.003 * v, 1e-4, 1e-6)
    want = O.leaky_relu((raw - m) / torch.sqrt(v + 1e-5) * gamma.cpu().double() + beta.cpu().double(), 0.1)
    check(act, want, 2.0 ** -6, 2e-2)


@pytest.mark.parametrize("B,H,W", [(2, 20, 24), (2, 9, 64), (1, 5, 96), (3, 33, 32), (1, 1, 32)])
def test_conv_first_layer(dev, B, H, W):
    """W % 32 == 0 runs the f32-MFMA kernel (32-pixel row tiles), other widths the thread-per-pixel kernel;
    both are exact-f32 products and sums, the output is rounded to bf16 once."""
    g = torch.Generator().manual_seed(3 + W)
    x = torch.rand(B, H, W, 3, generator=g)
    w = torch.randn(3, 3, 3, 32, generator=g) * 0.2
    scale = torch.rand(32, generator=g) + 0.5
    shift = torch.randn(32, generator=g) * 0.1
    want = O.leaky_relu(O.conv2d_same(x.double(), w.double(), 1) * scale.double() + shift.double(), 0.1)
    y = torch.empty(B, H, W, 32, dtype=torch.bfloat16, device=dev)
    L.conv_first_fwd(x.to(dev), w.to(dev), scale.to(dev), shift.to(dev), y)
    torch.cuda.synchronize()
    check(y, want, 2.0 ** -8, 1e-5)


@pytest.mark.parametrize("B,S", [(2, 64), (1, 96), (3, 32), (1, 160)])
def test_conv12_fused_matches_the_two_layers(dev, B, S):
    """conv1 + conv2 in one launch (inference-mode BN folded): against the f64 chain with conv1's output rounded to bf16
    where the unfused path stores it, and against the unfused kernels themselves (conv1 on hi/lo-split bf16 operands is
    2^-16 from exact f32, so a bf16 rounding of act1 flips now and then: act2 agrees bit for bit almost everywhere)"""
    g = torch.Generator().manual_seed(S + B)
    img = torch.rand(B, S, S, 3, generator=g)
    w1 = torch.randn(3, 3, 3, 32, generator=g) * 0.3
    w2 = bf16r(torch.randn(3, 3, 32, 64, generator=g) / 17)
    sc1, sh1 = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    sc2, sh2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    assert L.conv12_fused_ok(B, S, S)
    a1 = bf16r(O.leaky_relu(O.conv2d_same(img.double(), w1.double(), 1) * sc1.double() + sh1.double(), 0.1).float())
    want = O.leaky_relu(O.conv2d_same(a1.double(), w2.double(), 2) * sc2.double() + sh2.double(), 0.1)
    w2p = pack_ref(w2).to(torch.bfloat16).to(dev)
    y = torch.full((B, S // 2, S // 2, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    L.conv12_fused_fwd(img.to(dev), w1.to(dev), sc1.to(dev), sh1.to(dev), w2p, sc2.to(dev), sh2.to(dev), y, alpha=0.1)
    torch.cuda.synchronize()
    check(y, want, 2.0 ** -7, 3e-3)
    # the unfused kernels on the same operands
    y1 = torch.empty(B, S, S, 32, dtype=torch.bfloat16, device=dev)
    L.conv_first_fwd(img.to(dev), w1.to(dev), sc1.to(dev), sh1.to(dev), y1, alpha=0.1)
    y2 = torch.empty_like(y)
    L.conv2d_fwd(L.make_conv_desc(y1, w2p, y2, 3, 2, scale=sc2.to(dev), shift=sh2.to(dev), leaky=True))
    torch.cuda.synchronize()
    same = float((y.view(torch.int16) == y2.view(torch.int16)).float().mean())
    assert same > 0.97, same
    assert float((y.float() - y2.float()).abs().max()) <= 2.0 ** -6 * float(y2.float().abs().max())


def _block32_operands(B, H, W, C1, seed):
    g = torch.Generator().manual_seed(seed)
    x0 = bf16r(torch.randn(B, H, W, 64, generator=g))
    x1 = bf16r(torch.randn(B, H // 2, W // 2, C1, generator=g)) if C1 else None
    wA = bf16r(torch.randn(1, 1, 64 + C1, 32, generator=g) / 8)
    wB = bf16r(torch.randn(3, 3, 32, 64, generator=g) / 17)
    scA, shA = (torch.rand(32, generator=g) + 0.5), torch.randn(32, generator=g) * 0.2
    scB, shB = (torch.rand(64, generator=g) + 0.5), torch.randn(64, generator=g) * 0.2
    return x0, x1, wA, wB, scA, shA, scB, shB


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (1, 24, 48), (3, 8, 16), (1, 40, 16)])
def test_block32_fused_residual_block(dev, B, H, W):
    """[1x1 64 -> 32] -> [3x3 32 -> 64] + residual in one launch: against the f64 chain with the intermediate rounded to
    bf16 where the unfused path stores it, and against the two unfused kernels on the same operands (same products, the
    f32 sums in a different order: a bf16 rounding flips now and then)"""
    x0, _, wA, wB, scA, shA, scB, shB = _block32_operands(B, H, W, 0, H + W + B)
    assert L.block32_fused_ok(B, H, W, 64, 0, 0)
    a3 = bf16r(O.leaky_relu(O.conv2d_same(x0, wA, 1) * scA.double() + shA.double(), 0.1).float())
    want = O.leaky_relu(O.conv2d_same(a3, wB, 1) * scB.double() + shB.double(), 0.1) + x0
    xd = x0.to(torch.bfloat16).to(dev)
    wAp, wBp = pack_ref(wA).to(torch.bfloat16).to(dev), pack_ref(wB).to(torch.bfloat16).to(dev)
    y = torch.full((B, H, W, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    L.block32_fused_fwd(xd, None, wAp, scA.to(dev), shA.to(dev), wBp, scB.to(dev), shB.to(dev), y, post=0, alpha=0.1)
    torch.cuda.synchronize()
    check(y, want, 2.0 ** -7, 3e-3)
    y3 = torch.empty(B, H, W, 32, dtype=torch.bfloat16, device=dev)
    L.conv2d_fwd(L.make_conv_desc(xd, wAp, y3, 1, 1, scale=scA.to(dev), shift=shA.to(dev), leaky=True))
    y4 = torch.empty_like(y)
    L.conv2d_fwd(L.make_conv_desc(y3, wBp, y4, 3, 1, scale=scB.to(dev), shift=shB.to(dev), residual=xd, leaky=True))
    torch.cuda.synchronize()
    same = float((y.view(torch.int16) == y4.view(torch.int16)).float().mean())
    assert same > 0.97, same
    assert float((y.float() - y4.float()).abs().max()) <= 2.0 ** -6 * float(y4.float().abs().max())


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (1, 24, 48), (3, 8, 16), (1, 40, 16), (2, 128, 144)])
def test_block64_fused_residual_block(dev, B, H, W):
    """[1x1 128 -> 64] -> [3x3 64 -> 128] + residual in one launch (one persistent block per CU, input tile double-buffered;
    2 x 128 x 144 = 288 patches: 32 blocks walk over two; 40 x 16 and 24 x 48 put a patch on every border)"""
    g = torch.Generator().manual_seed(H + W + B + 7)
    x0 = bf16r(torch.randn(B, H, W, 128, generator=g))
    wA = bf16r(torch.randn(1, 1, 128, 64, generator=g) / 11)
    wB = bf16r(torch.randn(3, 3, 64, 128, generator=g) / 24)
    scA, shA = (torch.rand(64, generator=g) + 0.5), torch.randn(64, generator=g) * 0.2
    scB, shB = (torch.rand(128, generator=g) + 0.5), torch.randn(128, generator=g) * 0.2
    assert L.block64_fused_ok(B, H, W, 128) and not L.block64_fused_ok(B, H, W, 64)
    am = bf16r(O.leaky_relu(O.conv2d_same(x0, wA, 1) * scA.double() + shA.double(), 0.1).float())
    want = O.leaky_relu(O.conv2d_same(am, wB, 1) * scB.double() + shB.double(), 0.1) + x0
    xd = x0.to(torch.bfloat16).to(dev)
    wAp, wBp = pack_ref(wA).to(torch.bfloat16).to(dev), pack_ref(wB).to(torch.bfloat16).to(dev)
    y = torch.full((B, H, W, 128), float("nan"), dtype=torch.bfloat16, device=dev)
    L.block64_fused_fwd(xd, wAp, scA.to(dev), shA.to(dev), wBp, scB.to(dev), shB.to(dev), y, alpha=0.1)
    torch.cuda.synchronize()
    check(y, want, 2.0 ** -7, 3e-3)
    ym = torch.empty(B, H, W, 64, dtype=torch.bfloat16, device=dev)
    L.conv2d_fwd(L.make_conv_desc(xd, wAp, ym, 1, 1, scale=scA.to(dev), shift=shA.to(dev), leaky=True))
    y2 = torch.empty_like(y)
    L.conv2d_fwd(L.make_conv_desc(ym, wBp, y2, 3, 1, scale=scB.to(dev), shift=shB.to(dev), residual=xd, leaky=True))
    torch.cuda.synchronize()
    same = float((y.view(torch.int16) == y2.view(torch.int16)).float().mean())
    assert same > 0.97, same
    assert float((y.float() - y2.float()).abs().max()) <= 2.0 ** -6 * float(y2.float().abs().max())


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (1, 24, 48), (3, 8, 16)])
def test_block32_fused_mask_head(dev, B, H, W):
    """[1x1 (64 + up2(32)) -> 32] -> [3x3 32 -> 64] -> [1x1 64 -> 9] + bias in one launch, f32 out"""
    x0, x1, wA, wB, scA, shA, scB, shB = _block32_operands(B, H, W, 32, H + W + B + 1)
    g = torch.Generator().manual_seed(99)
    wC = bf16r(torch.randn(1, 1, 64, 9, generator=g) / 8)
    bC = torch.randn(9, generator=g) * 0.3
    assert L.block32_fused_ok(B, H, W, 64, 32, 1)
    up = x1.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    xin = torch.cat([x0, up], dim=3)
    a80 = bf16r(O.leaky_relu(O.conv2d_same(xin, wA, 1) * scA.double() + shA.double(), 0.1).float())
    a81 = bf16r(O.leaky_relu(O.conv2d_same(a80, wB, 1) * scB.double() + shB.double(), 0.1).float())
    want = O.conv2d_same(a81, wC, 1) + bC.double()
    x0d, x1d = x0.to(torch.bfloat16).to(dev), x1.to(torch.bfloat16).to(dev)
    wAp, wBp, wCp = (pack_ref(w).to(torch.bfloat16).to(dev) for w in (wA, wB, wC))
    y = torch.full((B, H, W, 9), float("nan"), dtype=torch.float32, device=dev)
    L.block32_fused_fwd(x0d, x1d, wAp, scA.to(dev), shA.to(dev), wBp, scB.to(dev), shB.to(dev), y, post=1, wC=wCp, biasC=bC.to(dev),
                        alpha=0.1)
    torch.cuda.synchronize()
    check(y, want, 2.0 ** -7, 2e-2)     # (a flipped bf16 rounding of act80 / act81 moves a sum of 64 products by ~1e-2)
    y80 = torch.empty(B, H, W, 32, dtype=torch.bfloat16, device=dev)
    L.conv2d_fwd(L.make_conv_desc(x0d, wAp, y80, 1, 1, x1=x1d, scale=scA.to(dev), shift=shA.to(dev), leaky=True))
    y81 = torch.empty(B, H, W, 64, dtype=torch.bfloat16, device=dev)
    L.conv2d_fwd(L.make_conv_desc(y80, wBp, y81, 3, 1, scale=scB.to(dev), shift=shB.to(dev), leaky=True))
    y82 = torch.empty_like(y)
    L.conv2d_fwd(L.make_conv_desc(y81, wCp, y82, 1, 1, shift=bC.to(dev), out_f32=True))
    torch.cuda.synchronize()
    assert float((y - y82).abs().max()) <= 2e-2 * max(1.0, float(y82.abs().max()))
    assert float(((y - y82).abs() <= 1e-5 * (1 + y82.abs())).float().mean()) > 0.9


def test_pack_weights(dev):
    g = torch.Generator().manual_seed(5)
    for k, cin, cout, pad in ((3, 64, 128, 128), (1, 96, 24, 32), (3, 32, 9, 32)):
        w = torch.randn(k, k, cin, cout, generator=g)
        wf = torch.empty(cout, k * k * cin, dtype=torch.bfloat16, device=dev)
        wdg = torch.empty(cin, k * k * pad, dtype=torch.bfloat16, device=dev)
        L.pack_weights(w.to(dev), wf, wdg, k, cin, cout, pad)
        torch.cuda.synchronize()
        assert torch.equal(wf.cpu(), pack_ref(w).to(torch.bfloat16))
        ref = torch.zeros(cin, k, k, pad)
        ref[..., :cout] = w.flip(0, 1).permute(2, 0, 1, 3)
        assert torch.equal(wdg.cpu(), ref.reshape(cin, -1).to(torch.bfloat16))


DGRAD = [(2, 18, 18, 64, 128, 3, 1), (2, 12, 12, 128, 64, 1, 1), (1, 20, 20, 32, 64, 3, 2), (2, 17, 19, 32, 64, 3, 2)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", DGRAD)
def test_conv_dgrad_matches_autograd(dev, B, H, W, Cin, Cout, k, s):
    """data gradient = forward kernel over dy with the flipped/transposed operand made by
    pack_weights, pads k-1-pad, and the transposed gather (in_div = stride)."""
    g = torch.Generator().manual_seed(Cin + Cout + k + s)
    x = torch.randn(B, H, W, Cin, generator=g, dtype=torch.float64, requires_grad=True)
    w = bf16r(torch.randn(k, k, Cin, Cout, generator=g) / (k * k * Cin) ** 0.5)
    y = O.conv2d_same(x, w, s)
    dy = bf16r(torch.randn(y.shape, generator=g))
    y.backward(dy)
    want = x.grad
    Ho, pt, _ = O.same_pads(H, k, s)
    Wo, pl, _ = O.same_pads(W, k, s)
    wdg = torch.empty(Cin, k * k * Cout, dtype=torch.bfloat16, device=dev)
    L.pack_weights(w.float().to(dev), None, wdg, k, Cin, Cout, Cout)
    dx = torch.empty(B, H, W, Cin, dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(dy.to(torch.bfloat16).to(dev), wdg, dx, k, 1, in_div=s, pads=(k - 1 - pt, k - 1 - pl),
                         out_hw=(H, W))
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    check(dx, want, 2.0 ** -7, 2e-3)


# stride-2 data gradients by output-parity classes (even output sizes; the GEMM tiles): several tiles per class, ragged class
# size, every tile shape, both BK, accumulation into an existing gradient
DGRAD_S2 = [(2, 36, 36, 32, 64, 0, False), (2, 36, 36, 32, 64, 2, True), (1, 40, 24, 64, 128, 12, False), (3, 20, 28, 64, 128, 3, True),
            (2, 48, 48, 32, 64, 4, False), (1, 72, 72, 32, 64, 0x206, True), (2, 22, 26, 128, 256, 0x10c, False), (2, 12, 16, 512, 1024, 3, False)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,tile,accumulate", DGRAD_S2)
def test_stride2_dgrad_by_parity_classes(dev, B, H, W, Cin, Cout, tile, accumulate):
    """dx of a 3x3 stride-2 conv: a tap reaches an input pixel only where the parities match, so the GEMM tiles are formed
    per output-parity class and skip the 5-8 void taps of their class.  Against autograd (f64), with and without a gradient
    already in dx (residual = dx: the second consumer of a tensor), and bit for bit against the all-taps order
    (DISYOLO_DGRAD_PCLS=0 is read once per process, so that comparison lives in tools/; here: values)."""
    g = torch.Generator().manual_seed(Cin + Cout + H + tile)
    x = torch.randn(B, H, W, Cin, generator=g, dtype=torch.float64, requires_grad=True)
    w = bf16r(torch.randn(3, 3, Cin, Cout, generator=g) / (9 * Cin) ** 0.5)
    y = O.conv2d_same(x, w, 2)
    dy = bf16r(torch.randn(y.shape, generator=g))
    y.backward(dy)
    prev = bf16r(torch.randn(B, H, W, Cin, generator=g)) if accumulate else None
    want = x.grad + (prev if accumulate else 0.0)
    _, pt, _ = O.same_pads(H, 3, 2)
    _, pl, _ = O.same_pads(W, 3, 2)
    wdg = torch.empty(Cin, 9 * Cout, dtype=torch.bfloat16, device=dev)
    L.pack_weights(w.float().to(dev), None, wdg, 3, Cin, Cout, Cout)
    dx = prev.to(torch.bfloat16).to(dev) if accumulate else torch.full((B, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(dy.to(torch.bfloat16).to(dev), wdg, dx, 3, 1, in_div=2, pads=(2 - pt, 2 - pl), out_hw=(H, W),
                         residual=dx if accumulate else None, tile=tile)
    L.conv2d_fwd(d)
    torch.cuda.synchronize()
    check(dx, want, 2.0 ** -7, 2e-3)


@pytest.mark.parametrize("B,H,W,C,Cdy,accumulate", [(2, 36, 36, 32, 64, False), (1, 40, 24, 64, 128, True), (3, 20, 28, 32, 64, True),
                                                     (2, 64, 48, 32, 64, False), (1, 16, 16, 64, 128, False)])
def test_stride2_dgrad_quad(dev, B, H, W, C, Cdy, accumulate):
    """the shallow stride-2 layers' data gradient as one 2x2-tap conv over dy with a depth-to-space store (pack_quad +
    dgrad_s2_quad) against autograd, and against the generic in_div = 2 path (same products, other summation order)"""
    g = torch.Generator().manual_seed(C + Cdy + H)
    x = torch.randn(B, H, W, C, generator=g, dtype=torch.float64, requires_grad=True)
    w = bf16r(torch.randn(3, 3, C, Cdy, generator=g) / (9 * C) ** 0.5)
    y = O.conv2d_same(x, w, 2)
    dy = bf16r(torch.randn(y.shape, generator=g))
    y.backward(dy)
    prev = bf16r(torch.randn(B, H, W, C, generator=g)) if accumulate else None
    want = x.grad + (prev if accumulate else 0.0)
    assert L.dgrad_s2_quad_ok(B, H // 2, W // 2, Cdy, C)
    wq = torch.full((4 * C, 9 * Cdy), float("nan"), dtype=torch.bfloat16, device=dev)
    L.pack_quad(w.float().to(dev), wq)
    dyd = dy.to(torch.bfloat16).to(dev)
    dx = prev.to(torch.bfloat16).to(dev) if accumulate else torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device=dev)
    L.dgrad_s2_quad(dyd, wq, dx, accumulate=accumulate)
    torch.cuda.synchronize()
    assert torch.isfinite(wq.float()).all()
    check(dx, want, 2.0 ** -7, 2e-3)
    wdg = torch.empty(C, 9 * Cdy, dtype=torch.bfloat16, device=dev)
    L.pack_weights(w.float().to(dev), None, wdg, 3, C, Cdy, Cdy)
    dx2 = prev.to(torch.bfloat16).to(dev) if accumulate else torch.empty(B, H, W, C, dtype=torch.bfloat16, device=dev)
    L.conv2d_fwd(L.make_conv_desc(dyd, wdg, dx2, 3, 1, in_div=2, pads=(2, 2), out_hw=(H, W), residual=dx2 if accumulate else None))
    torch.cuda.synchronize()
    assert float((dx.float() - dx2.float()).abs().max()) <= 2.0 ** -6 * float(dx2.float().abs().max())


WGRAD = [(2, 18, 18, 64, 128, 3, 1), (2, 12, 12, 128, 64, 1, 1), (1, 20, 20, 32, 64, 3, 2), (2, 18, 18, 256, 24, 1, 1),
         (1, 24, 24, 64, 9, 1, 1), (3, 10, 10, 96, 32, 1, 1), (2, 36, 36, 32, 64, 3, 1),
         # the tap-fused 3x3 kernel: every ring size (W+1 = 73 / 145 / 289 / 421), both channel tiles, batches
         # crossing a chunk, H != W, frames smaller than one 64-pixel chunk, Cout not a multiple of 16
         (1, 72, 72, 32, 128, 3, 1), (1, 144, 144, 32, 64, 3, 1), (1, 288, 288, 32, 32, 3, 1),
         (1, 40, 420, 32, 32, 3, 1), (5, 2, 2, 32, 64, 3, 1), (3, 5, 7, 64, 36, 3, 1), (8, 18, 18, 64, 256, 3, 1),
         (2, 26, 26, 96, 192, 3, 1),
         # the same kernel on the stride-2 layers (four parity-class rings): conv2's full-size map (289-pixel frame rows, all
         # eight ring slots in use), conv5 / conv10 shapes, H != W, a frame smaller than a chunk, Cout not a multiple of 16
         (1, 576, 576, 32, 64, 3, 2), (2, 144, 144, 64, 128, 3, 2), (2, 36, 36, 128, 256, 3, 2), (3, 36, 28, 96, 36, 3, 2),
         (5, 4, 6, 32, 64, 3, 2), (1, 72, 200, 32, 32, 3, 2),
         # an output row too long for four 8-chunk rings (351-pixel frame rows): the planner falls back to im2col
         (1, 6, 700, 32, 32, 3, 2)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s", WGRAD)
def test_conv_wgrad_matches_autograd(dev, B, H, W, Cin, Cout, k, s):
    g = torch.Generator().manual_seed(Cin * 3 + Cout + k + s)
    x = bf16r(torch.randn(B, H, W, Cin, generator=g))
    w = torch.randn(k, k, Cin, Cout, generator=g, dtype=torch.float64, requires_grad=True)
    y = O.conv2d_same(x, w, s)
    dy = bf16r(torch.randn(y.shape, generator=g))
    y.backward(dy)
    want = w.grad
    ld = ((Cout + 31) // 32) * 32 if Cout % 8 else Cout
    dyp = torch.zeros(y.shape[0], y.shape[1], y.shape[2], ld, dtype=torch.bfloat16, device=dev)
    dyp[..., :Cout] = dy.to(torch.bfloat16).to(dev)
    dw = torch.full((k, k, Cin, Cout), float("nan"), dtype=torch.float32, device=dev)
    dummy = torch.empty(1, dtype=torch.bfloat16, device=dev)
    yd = torch.empty(y.shape, dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(x.to(torch.bfloat16).to(dev), dummy, yd, k, s)
    if k == 3 and s == 2 and Cin % 32 == 0 and Cout >= 32 and W // 2 + 1 <= 320:
        assert L.conv2d_wgrad_plan(d)[0] == 1          # the tap-fused kernel, not im2col
    if k == 3 and s == 2 and W // 2 + 1 > 320:
        assert L.conv2d_wgrad_plan(d)[0] == 0
    L.conv2d_wgrad(d, dyp, ld, dw, L.Workspace(dev))
    torch.cuda.synchronize()
    check(dw, want, 1e-4, 2e-4 * float(want.abs().max()))


@pytest.mark.parametrize("B,H,W", [(1, 576, 576), (2, 64, 200), (3, 7, 5), (1, 130, 129), (2, 33, 128), (1, 1, 1)])
def test_first_layer_wgrad_on_the_matrix_cores(dev, B, H, W):
    """conv1's weight gradient (3 -> 32 filters): taps as the M axis of the MFMA, 128-pixel row units, image rounded to bf16
    on the way into LDS -- against autograd on the same bf16-rounded image; ragged rows (W not a multiple of 128, one
    pixel past a unit), maps smaller than a unit, the full-size map"""
    g = torch.Generator().manual_seed(B * 7 + H + W)
    img = torch.rand(B, H, W, 3, generator=g) * 2 - 1
    w = torch.randn(3, 3, 3, 32, generator=g, dtype=torch.float64, requires_grad=True)
    y = O.conv2d_same(bf16r(img), w, 1)
    dy = bf16r(torch.randn(y.shape, generator=g))
    y.backward(dy)
    want = w.grad
    dw = torch.full((3, 3, 3, 32), float("nan"), dtype=torch.float32, device=dev)
    L.conv_first_wgrad(img.to(dev), dy.to(torch.bfloat16).to(dev), dw, L.Workspace(dev))
    torch.cuda.synchronize()
    check(dw, want, 1e-4, 2e-4 * float(want.abs().max()))


def test_wgrad_fused_concat(dev):
    g = torch.Generator().manual_seed(9)
    skip = bf16r(torch.randn(2, 12, 12, 64, generator=g))
    low = bf16r(torch.randn(2, 6, 6, 32, generator=g))
    w = torch.randn(1, 1, 96, 32, generator=g, dtype=torch.float64, requires_grad=True)
    cat = torch.cat([skip, O.upsample2(low)], dim=-1)
    y = O.conv2d_same(cat, w, 1)
    dy = bf16r(torch.randn(y.shape, generator=g))
    y.backward(dy)
    dw = torch.empty(1, 1, 96, 32, dtype=torch.float32, device=dev)
    dummy = torch.empty(1, dtype=torch.bfloat16, device=dev)
    yd = torch.empty(y.shape, dtype=torch.bfloat16, device=dev)
    d = L.make_conv_desc(skip.to(torch.bfloat16).to(dev), dummy, yd, 1, 1, x1=low.to(torch.bfloat16).to(dev))
    L.conv2d_wgrad(d, dy.to(torch.bfloat16).to(dev), 32, dw, L.Workspace(dev))
    torch.cuda.synchronize()
    check(dw, w.grad, 1e-4, 2e-4 * float(w.grad.abs().max()))


BN_BWD_CASES = [
    # B, H, W, Cin, Cout, tile, residual
    (2, 18, 18, 64, 128, 16, True),        # patch kernel, 8 waves
    (2, 18, 36, 64, 72, 17, False),        # 4 waves, ragged channel tile (72 = 64 + 8)
    (2, 18, 18, 96, 64, 18, True),         # 32 channels per block: one write-out round
    (2, 18, 18, 96, 64, 19, True),         # 16 channels per block
    (3, 9, 9, 256, 256, 16, False),        # 9x9 patch = whole image, 6 of 24 fragment slots used
    (1, 36, 36, 64, 64, 16, True),         # four patches per image
    (2, 18, 18, 64, 128, 24, True),        # flat-frame kernel, 192 x 64 tiles: every K group finishes a fragment
    (8, 18, 18, 128, 64, 24, False),       # ... 16 M tiles
    (2, 36, 36, 64, 128, 25, True),        # 384 x 64 tiles: K group g finishes channel half g
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,tile,with_res", BN_BWD_CASES)
def test_patch_conv_emits_batchnorm_backward_sums(dev, B, H, W, Cin, Cout, tile, with_res):
    """DISYOLO_CONV_BN_BWD_STATS: a patch-kernel conv whose output is the final gradient wrt a batch-normalised
    layer's output also writes, per patch, that layer's (sum g, sum g*xhat) -- computed from the bf16 values it
    stores, so summed over the patches they equal the column reduction over the stored tensor (f32 sums: 1e-4 of
    the column's sum of magnitudes).  The output itself is unchanged by the flag, bit for bit."""
    g = torch.Generator().manual_seed(B + H + Cin + Cout + tile)
    x = bf16r(torch.randn(B, H, W, Cin, generator=g))
    w = bf16r(torch.randn(3, 3, Cin, Cout, generator=g) / (9 * Cin) ** 0.5)
    res = bf16r(torch.randn(B, H, W, Cout, generator=g)) if with_res else None
    bn_x = bf16r(torch.randn(B, H, W, Cout, generator=g) * 2 + 0.3)
    mean = torch.randn(Cout, generator=g) * 0.2 + 0.3
    rstd = torch.rand(Cout, generator=g) + 0.3
    scale = torch.randn(Cout, generator=g) * rstd
    shift = torch.randn(Cout, generator=g) * 0.5 - mean * scale
    xd, wd = x.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev)
    resd = res.to(torch.bfloat16).to(dev) if with_res else None
    y0 = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    y1 = torch.empty_like(y0)
    plain = L.make_conv_desc(xd, wd, y0, 3, 1, residual=resd, tile=tile)
    assert L.conv2d_bn_bwd_stats_ok(plain)
    assert not L.conv2d_bn_bwd_stats_ok(L.make_conv_desc(xd, wd, y0, 3, 1, residual=resd, tile=3))      # a GEMM tile cannot
    L.conv2d_fwd(plain)
    rows = L.conv2d_stats_rows(plain)
    part = torch.full((rows, Cout, 2), float("nan"), device=dev)
    dev_t = [t.to(dev) for t in (scale, shift, mean, rstd)]
    L.conv2d_fwd(L.make_conv_desc(xd, wd, y1, 3, 1, residual=resd, tile=tile,
                                  bn_bwd=(bn_x.to(torch.bfloat16).to(dev), dev_t[0], dev_t[1], dev_t[2], dev_t[3], part, 0.1)))
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.isfinite(part).all(), "a partial-sum row was not written"
    dy = y1.double().cpu().reshape(-1, Cout)
    vx = bn_x.reshape(-1, Cout)
    z = vx * scale.double() + shift.double()
    gg = dy * torch.where(z > 0, 1.0, 0.1)
    xh = (vx - mean.double()) * rstd.double()
    got = part.double().sum(0).cpu()
    for q, term in ((0, gg), (1, gg * xh)):
        want = term.sum(0)
        tol = 1e-4 * term.abs().sum(0) + 1e-6
        assert ((got[:, q] - want).abs() <= tol).all(), (q, float((got[:, q] - want).abs().max()))
    with pytest.raises(L.DisyoloError):       # the flag on a kernel without that epilogue is refused, not ignored
        L.conv2d_fwd(L.make_conv_desc(xd, wd, y1, 3, 1, residual=resd, tile=3,
                                      bn_bwd=(bn_x.to(torch.bfloat16).to(dev), dev_t[0], dev_t[1], dev_t[2], dev_t[3], part, 0.1)))


BN_BWD_GEMM_CASES = [
    # B, H, W, Cin, Cout, k, tile, residual
    (2, 18, 18, 128, 64, 1, 6, True),        # 64x64
    (2, 18, 18, 128, 256, 1, 3, False),      # 64x128
    (2, 18, 18, 128, 256, 1, 0x203, True),   # ... alternative pipeline depth
    (2, 18, 18, 128, 64, 1, 2, True),        # 128x64
    (2, 18, 18, 256, 128, 1, 1, False),      # 128x128
    (2, 18, 18, 128, 128, 1, 10, True),      # 96x128: 648 pixels = 6.75 tiles (ragged last tile)
    (2, 18, 18, 128, 128, 1, 11, False),     # 96x128, deep pipeline
    (2, 18, 18, 128, 128, 1, 9, False),      # 64x128, deep pipeline
    (2, 18, 18, 128, 72, 1, 6, True),        # ragged channel tile (72 = 64 + 8)
    (2, 18, 18, 96, 128, 1, 3, False),       # BK = 32 (96 channels)
    (3, 18, 18, 128, 256, 1, 12, True),      # 192x128 (972 pixels: 5.06 tiles)
    (3, 18, 18, 128, 256, 1, 0x20c, False),  # 192x128, three stages
    (2, 18, 18, 64, 128, 3, 12, True),       # 192x128 on a 3x3 layer
    (1, 36, 36, 64, 256, 3, 0x20c, False),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,tile,with_res", BN_BWD_GEMM_CASES)
def test_gemm_tile_conv_emits_batchnorm_backward_sums(dev, monkeypatch, B, H, W, Cin, Cout, k, tile, with_res):
    """round 6: the GEMM tiles that carry the batch-norm backward epilogue (the EPI = 1 instances, conv_igemm.hip launch_ks
    HAS_BWD) can emit DISYOLO_CONV_BN_BWD_STATS rows too, one per pixel tile -- what the data gradients of the 1x1 layers run, whose
    targets' batch-norm backward needs a column reduction over (dy, x).  Opt-in (DISYOLO_BN_BWD_STATS_GEMM=1: measured slower in
    the step, profiles/r06_bn_inkernel.txt).  Same statement as for the patch kernels: output unchanged bit for bit, rows summed =
    the column reduction over the stored bf16 tensor."""
    monkeypatch.setenv("DISYOLO_BN_BWD_STATS_GEMM", "1")
    g = torch.Generator().manual_seed(B + H + Cin + Cout + tile + k)
    x = bf16r(torch.randn(B, H, W, Cin, generator=g))
    w = bf16r(torch.randn(k, k, Cin, Cout, generator=g) / (k * k * Cin) ** 0.5)
    res = bf16r(torch.randn(B, H, W, Cout, generator=g)) if with_res else None
    bn_x = bf16r(torch.randn(B, H, W, Cout, generator=g) * 2 + 0.3)
    mean = torch.randn(Cout, generator=g) * 0.2 + 0.3
    rstd = torch.rand(Cout, generator=g) + 0.3
    scale = torch.randn(Cout, generator=g) * rstd
    shift = torch.randn(Cout, generator=g) * 0.5 - mean * scale
    xd, wd = x.to(torch.bfloat16).to(dev), pack_ref(w).to(torch.bfloat16).to(dev)
    resd = res.to(torch.bfloat16).to(dev) if with_res else None
    y0 = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    y1 = torch.full_like(y0, float("nan"))
    plain = L.make_conv_desc(xd, wd, y0, k, 1, residual=resd, tile=tile)
    assert L.conv2d_bn_bwd_stats_ok(plain)
    L.conv2d_fwd(plain)
    rows = L.conv2d_stats_rows(plain)
    part = torch.full((rows, Cout, 2), float("nan"), device=dev)
    dev_t = [t.to(dev) for t in (scale, shift, mean, rstd)]
    L.conv2d_fwd(L.make_conv_desc(xd, wd, y1, k, 1, residual=resd, tile=tile,
                                  bn_bwd=(bn_x.to(torch.bfloat16).to(dev), dev_t[0], dev_t[1], dev_t[2], dev_t[3], part, 0.1)))
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.isfinite(part).all(), "a partial-sum row was not written"
    dy = y1.double().cpu().reshape(-1, Cout)
    vx = bn_x.reshape(-1, Cout)
    z = vx * scale.double() + shift.double()
    gg = dy * torch.where(z > 0, 1.0, 0.1)
    xh = (vx - mean.double()) * rstd.double()
    got = part.double().sum(0).cpu()
    for q, term in ((0, gg), (1, gg * xh)):
        want = term.sum(0)
        tol = 1e-4 * term.abs().sum(0) + 1e-6
        assert ((got[:, q] - want).abs() <= tol).all(), (q, float((got[:, q] - want).abs().max()))


def test_gemm_tiles_without_the_epilogue_refuse_the_flag(dev, monkeypatch):
    """a 3x3 conv on a tile other than 192x128 and the narrow tiles have no such epilogue, and no GEMM tile has it unless
    DISYOLO_BN_BWD_STATS_GEMM=1: disyolo_conv2d_bn_bwd_stats_ok says so and the call with the flag raises"""
    x = torch.zeros(2, 18, 18, 128, dtype=torch.bfloat16, device=dev)
    assert not L.conv2d_bn_bwd_stats_ok(L.make_conv_desc(x, torch.zeros(128, 128, dtype=torch.bfloat16, device=dev), torch.zeros_like(x), 1, 1, tile=6))
    monkeypatch.setenv("DISYOLO_BN_BWD_STATS_GEMM", "1")
    x = torch.zeros(2, 18, 18, 128, dtype=torch.bfloat16, device=dev)
    y = torch.zeros(2, 18, 18, 128, dtype=torch.bfloat16, device=dev)
    w3 = torch.zeros(128, 9 * 128, dtype=torch.bfloat16, device=dev)
    w1 = torch.zeros(128, 128, dtype=torch.bfloat16, device=dev)
    assert not L.conv2d_bn_bwd_stats_ok(L.make_conv_desc(x, w3, y, 3, 1, tile=1))
    assert not L.conv2d_bn_bwd_stats_ok(L.make_conv_desc(x, w1, y, 1, 1, tile=4))          # 128x32
    assert L.conv2d_bn_bwd_stats_ok(L.make_conv_desc(x, w1, y, 1, 1, tile=6))
    v = torch.zeros(128, device=dev)
    part = torch.zeros(64, 128, 2, device=dev)
    with pytest.raises(L.DisyoloError):
        L.conv2d_fwd(L.make_conv_desc(x, w1, y, 1, 1, tile=4, bn_bwd=(y, v, v, v, v, part, 0.1)))


def test_bn_act_bwd_from_conv_partials_matches_the_plain_path(dev):
    """bn_act_bwd_partials(partials of the patch conv) == bn_act_bwd(column reduction over dy, x) up to the f32
    summation order of the two reductions (dgamma / dbeta to 2e-5 of the sum of magnitudes; dx within one bf16 ulp)."""
    B, H, W, Cin, Cout = 4, 18, 18, 128, 256
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, 9 * Cin, generator=g) / (9 * Cin) ** 0.5).to(torch.bfloat16).to(dev)
    raw = (torch.randn(B, H, W, Cout, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    mean = raw.float().reshape(-1, Cout).mean(0)
    rstd = 1.0 / torch.sqrt(raw.float().reshape(-1, Cout).var(0, unbiased=False) + 1e-5)
    scale = torch.randn(Cout, generator=g).to(dev) * rstd
    shift = (torch.randn(Cout, generator=g) * 0.3).to(dev) - mean * scale
    M = B * H * W
    ws = L.Workspace(dev)
    outs = []
    for fused in (False, True):
        dy = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
        dx = torch.empty_like(dy)
        dgamma, dbeta = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
        if fused:
            probe = L.make_conv_desc(x, w, dy, 3, 1, tile=16)
            rows = L.conv2d_stats_rows(probe)
            part = torch.empty(rows, Cout, 2, device=dev)
            L.conv2d_fwd(L.make_conv_desc(x, w, dy, 3, 1, tile=16, bn_bwd=(raw, scale, shift, mean, rstd, part, 0.1)))
            L.bn_act_bwd_partials(dy, raw, scale, shift, mean, rstd, dx, dgamma, dbeta, M, Cout, part, rows, ws, 0.1)
        else:
            L.conv2d_fwd(L.make_conv_desc(x, w, dy, 3, 1, tile=16))
            L.bn_act_bwd(dy, raw, scale, shift, mean, rstd, dx, dgamma, dbeta, M, Cout, ws, 0.1)
        torch.cuda.synchronize()
        outs.append((dx.float().cpu(), dgamma.cpu(), dbeta.cpu(), dy.float().cpu()))
    assert torch.equal(outs[0][3], outs[1][3])
    mag = outs[0][3].abs().reshape(-1, Cout).sum(0)
    assert ((outs[0][2] - outs[1][2]).abs() <= 2e-5 * mag + 1e-6).all()
    assert ((outs[0][1] - outs[1][1]).abs() <= 2e-5 * mag * 4 + 1e-6).all()
    check(outs[1][0], outs[0][0], 2.0 ** -7, 1e-4)

