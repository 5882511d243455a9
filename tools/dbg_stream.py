import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, numpy as np
import disyolo_amd
from disyolo_amd import lib as L
import disyolo_oracle as O
dev = torch.device("cuda:0")
B, H, W, Cout = 2, 36, 36, 64
g = torch.Generator().manual_seed(1)
x = torch.randint(-2, 3, (B, H, W, 32), generator=g).float()
w = torch.randint(-1, 2, (3, 3, 32, Cout), generator=g).float()
res = torch.randint(-3, 4, (B, H, W, Cout), generator=g).float()
wd = w.permute(3, 0, 1, 2).reshape(Cout, -1).contiguous().to(torch.bfloat16).to(dev)
xd = x.to(torch.bfloat16).to(dev)
conv = O.conv2d_same(x.double(), w.double(), 1)
for name, kw in (("plain", {}), ("res", dict(residual=res.to(torch.bfloat16).to(dev))), ("scale", dict(scale=torch.full((Cout,), 2.0, device=dev), shift=torch.full((Cout,), 1.0, device=dev)))):
    y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
    L.conv2d_fwd(L.make_conv_desc(xd, wd, y, 3, 1, tile=20, **kw))
    torch.cuda.synchronize()
    want = conv.clone()
    if "residual" in kw: want = want + res.double()
    if "scale" in kw: want = want * 2 + 1
    want = want.to(torch.bfloat16).double()
    bad = (y.float().cpu().double() != want)
    idx = bad.nonzero()
    print(name, "bad", int(bad.sum()), "of", bad.numel())
    if len(idx):
        print(" first", idx[:6].tolist(), "got", y.float().cpu()[tuple(idx[0])].item(), "want", want[tuple(idx[0])].item())
        print(" by y%18,x%18:", sorted(set((int(i[1]) % 18, int(i[2]) % 18) for i in idx))[:12], " channels:", sorted(set(int(i[3]) for i in idx))[:8], len(set(int(i[3]) for i in idx)))
