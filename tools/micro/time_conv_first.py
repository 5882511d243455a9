import sys, time
sys.path.insert(0, ".")
import torch, bench
from disyolo_amd import lib as L
dev = torch.device("cuda:0")
B, S = 8, 576
img = torch.rand(B, S, S, 3, device=dev)
w = torch.randn(3, 3, 3, 32, device=dev) * 0.1
sc = torch.rand(32, device=dev) + 0.5
sh = torch.randn(32, device=dev) * 0.1
y = torch.empty(B, S, S, 32, dtype=torch.bfloat16, device=dev)
for _ in range(5):
    L.conv_first_fwd(img, w, sc, sh, y, alpha=0.1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    L.conv_first_fwd(img, w, sc, sh, y, alpha=0.1)
e1.record(); torch.cuda.synchronize()
print("conv_first %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
