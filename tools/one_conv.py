"""Run ONE conv shape repeatedly (for rocprofv3 --pmc): python tools/one_conv.py H Cin Cout k s tile [B] [iters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
H, cin, cout, k, s = (int(v) for v in sys.argv[1:6])
tile = int(sys.argv[6], 0)
B = int(sys.argv[7]) if len(sys.argv) > 7 else 8
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
dev = torch.device("cuda:0")
Ho, _ = L.same_pads(H, k, s)
x0 = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, k * k * cin, device=dev) * 0.05).to(torch.bfloat16)
y = torch.empty(B, Ho, Ho, cout, dtype=torch.bfloat16, device=dev)
sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
d = L.make_conv_desc(x0, w, y, k, s, scale=sc, shift=sh, leaky=True, tile=tile)
probe = int(os.environ.get("PROBE", "0"), 0)
d.flags |= probe
import time
for _ in range(3): L.conv2d_fwd(d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): L.conv2d_fwd(d)
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) / 20 * 1e-3
print("probe %#x: %.1f us  %.1f TFLOP/s" % (probe, dt * 1e6, 2.0 * B * Ho * Ho * cout * cin * k * k / dt / 1e12))
for _ in range(iters):
    L.conv2d_fwd(d)
torch.cuda.synchronize()
print("done")
