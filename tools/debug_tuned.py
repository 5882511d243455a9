import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import disyolo_amd
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
import disyolo_oracle as O
dev = torch.device("cuda:0")
B, S = 2, 64
b = O.synthetic_batch(B, S, seed=33)
ref = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=8)
tuned = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=8)
tuned.load_state_dict(ref.state_dict())
for n in (ref, tuned):
    n.set_batch(b)
picks = tuned.autotune(reps=1, det_thresh=0.1)
ref._forward_layers(True); tuned._forward_layers(True)
torch.cuda.synchronize()
for l0, l1 in zip(ref.layers, tuned.layers):
    a, c = l0.act.float(), l1.act.float()
    d = (a - c).abs().max().item()
    key = L.conv_shape_key(l1.desc) if getattr(l1, "desc", None) is not None else None
    extra = ""
    if l1.raw is not None and l0.raw is not None:
        extra = " raw %.3g" % (l0.raw.float() - l1.raw.float()).abs().max().item()
    print(l0.idx, "tile", hex(picks.get(key, 0)) if key else "-", "lock", l0.lock, "maxdiff %.4g of %.3g" % (d, a.abs().max().item()), extra)
