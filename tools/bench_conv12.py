"""conv1 + conv2: the fused launch against the two separate kernels (GPU box): python tools/bench_conv12.py [B] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 576
dev = torch.device("cuda:0")
img = torch.rand(B, S, S, 3, device=dev)
w1 = torch.randn(3, 3, 3, 32, device=dev) * 0.3
w2p = (torch.randn(64, 288, device=dev) / 17).to(torch.bfloat16)
sc1, sh1 = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.2
sc2, sh2 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2
y1 = torch.empty(B, S, S, 32, dtype=torch.bfloat16, device=dev)
y2 = torch.empty(B, S // 2, S // 2, 64, dtype=torch.bfloat16, device=dev)
d2 = L.make_conv_desc(y1, w2p, y2, 3, 2, scale=sc2, shift=sh2, leaky=True)
flush = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        flush.fill_(1)                      # evict the Infinity Cache: these layers run on cold data in the step
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / n * 1e3
t1 = timeit(lambda: L.conv_first_fwd(img, w1, sc1, sh1, y1, alpha=0.1))
t2 = timeit(lambda: L.conv2d_fwd(d2))
tf = timeit(lambda: L.conv12_fused_fwd(img, w1, sc1, sh1, w2p, sc2, sh2, y2, alpha=0.1))
mb = (B * S * S * 12 + B * (S // 2) ** 2 * 128) / 1e6
print("B=%d S=%d: conv1 %.1f us + conv2 %.1f us = %.1f us; fused %.1f us (%.0f MB -> %.2f TB/s)" % (B, S, t1, t2, t1 + t2, tf, mb, mb / tf))
