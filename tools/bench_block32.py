"""The fused [1x1 -> 32] -> [3x3 32 -> 64] launches against the separate kernels (GPU box): python tools/bench_block32.py [B] [S]
S = the map size (288 for the 576^2 network)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 288
dev = torch.device("cuda:0")
bf = torch.bfloat16
x0 = torch.randn(B, S, S, 64, device=dev).to(bf)
x1 = torch.randn(B, S // 2, S // 2, 32, device=dev).to(bf)
wA0 = (torch.randn(32, 64, device=dev) / 8).to(bf)
wA1 = (torch.randn(32, 96, device=dev) / 10).to(bf)
wB = (torch.randn(64, 288, device=dev) / 17).to(bf)
wC = (torch.randn(9, 64, device=dev) / 8).to(bf)
scA, shA = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.2
scB, shB = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2
bC = torch.randn(9, device=dev)
y3 = torch.empty(B, S, S, 32, dtype=bf, device=dev)
y4 = torch.empty(B, S, S, 64, dtype=bf, device=dev)
y9 = torch.empty(B, S, S, 9, dtype=torch.float32, device=dev)
flush = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
tune = {"3": (21, 3, 6), "4": (20, 2, 0x202), "80": (21, 4), "82": (21, 4)}
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        flush.fill_(1)                      # evict the Infinity Cache: these layers run on cold data in the step
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / n * 1e3
def best(mk, tiles):
    return min(timeit(lambda d=mk(t): L.conv2d_fwd(d)) for t in tiles)
t3 = best(lambda t: L.make_conv_desc(x0, wA0, y3, 1, 1, scale=scA, shift=shA, leaky=True, tile=t), (21, 6, 0x206))
t4 = best(lambda t: L.make_conv_desc(y3, wB, y4, 3, 1, scale=scB, shift=shB, residual=x0, leaky=True, tile=t), (20, 2, 0x202))
tf = timeit(lambda: L.block32_fused_fwd(x0, None, wA0, scA, shA, wB, scB, shB, y4, post=0))
mb = B * S * S * 256 / 1e6
print("B=%d %d^2 residual block: 1x1 %.1f us + 3x3 %.1f us = %.1f us; fused %.1f us (%.0f MB -> %.2f TB/s)" % (B, S, t3, t4, t3 + t4, tf, mb, mb / tf))
t80 = best(lambda t: L.make_conv_desc(x0, wA1, y3, 1, 1, x1=x1, scale=scA, shift=shA, leaky=True, tile=t), (21, 4, 0x204))
t81 = best(lambda t: L.make_conv_desc(y3, wB, y4, 3, 1, scale=scB, shift=shB, leaky=True, tile=t), (20, 2, 0x202))
t82 = best(lambda t: L.make_conv_desc(y4, wC, y9, 1, 1, shift=bC, out_f32=True, tile=t), (21, 4, 0x204))
tf = timeit(lambda: L.block32_fused_fwd(x0, x1, wA1, scA, shA, wB, scB, shB, y9, post=1, wC=wC, biasC=bC))
mb = (B * S * S * (128 + 36) + B * (S // 2) ** 2 * 64) / 1e6
print("B=%d %d^2 mask head: %.1f + %.1f + %.1f = %.1f us; fused %.1f us (%.0f MB -> %.2f TB/s)" % (B, S, t80, t81, t82, t80 + t81 + t82, tf, mb, mb / tf))

# the 144^2 residual blocks (conv6+7, conv8+9): [1x1 128 -> 64] -> [3x3 64 -> 128] + residual
S2 = S // 2
xa = torch.randn(B, S2, S2, 128, device=dev).to(bf)
wa = (torch.randn(64, 128, device=dev) / 11).to(bf)
wb = (torch.randn(128, 576, device=dev) / 24).to(bf)
sa, ha = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2
sb, hb = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.2
ym = torch.empty(B, S2, S2, 64, dtype=bf, device=dev)
yo = torch.empty(B, S2, S2, 128, dtype=bf, device=dev)
t6 = best(lambda t: L.make_conv_desc(xa, wa, ym, 1, 1, scale=sa, shift=ha, leaky=True, tile=t), (21, 6, 0x206, 2))
t7 = best(lambda t: L.make_conv_desc(ym, wb, yo, 3, 1, scale=sb, shift=hb, residual=xa, leaky=True, tile=t), (0, 16, 12, 2, 0x202, 3))
tf = timeit(lambda: L.block64_fused_fwd(xa, wa, sa, ha, wb, sb, hb, yo))
mb = B * S2 * S2 * 512 / 1e6
gf = 2.0 * B * S2 * S2 * (64 * 128 + 128 * 576) / 1e9
print("B=%d %d^2 residual block 128: 1x1 %.1f us + 3x3 %.1f us = %.1f us; fused %.1f us (%.0f MB -> %.2f TB/s, %.0f TFLOP/s)"
      % (B, S2, t6, t7, t6 + t7, tf, mb, mb / tf, gf / tf * 1e3))
