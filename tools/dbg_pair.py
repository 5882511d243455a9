import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import torch, numpy as np
import disyolo_amd
from disyolo_amd.net import YOLONet
import disyolo_oracle as O
dev = torch.device("cuda:0")
B, S = 2, 64
batches = [O.synthetic_batch(B, S, seed=90 + t) for t in range(4)]
val = O.synthetic_batch(B, S, seed=99)
pair = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=8, backbone_pair=True)
plain = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=8)
for t in range(2):
    pair.train_step((batches[2 * t], batches[2 * t + 1]) if t % 2 == 0 else None, det_thresh=0.1)
    plain.load_state_dict(pair.state_dict())
    pair.forward(val["images"], val["clip_window"], [0.1], is_training=False)
    plain.forward(val["images"], val["clip_window"], [0.1], is_training=False)
    torch.cuda.synchronize()
    print("t", t, "half", pair._half)
    for l in plain.layers:
        a = pair.by_idx[l.idx].act
        if l.idx <= pair._pair_P:
            a = a[:B]
        b = l.act
        e = float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        if e > 1e-3 or l.idx in (52, 53, 59, 75, 82):
            print("  layer", l.idx, "rel", e, "norm", float(b.double().norm()))
