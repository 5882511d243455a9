"""how long are the per-class candidate lists of the detection filter in the bench workloads (random-init weights)?
(python tools/count_candidates.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import disyolo_amd
from disyolo_amd import lib as L, config as cfg
from disyolo_amd.net import YOLONet
from bench import synthetic_batch
dev = torch.device("cuda:0")


def lists(net, B, S):
    NC = 3 * sum((S // d) ** 2 for d in (8, 16, 32))
    raw = net.ws_det.buf
    sc = raw[B * NC * 16:B * NC * 20].view(torch.float32).reshape(B, NC)
    cl = raw[B * NC * 20:B * NC * 24].view(torch.int32).reshape(B, NC)
    ok = sc > cfg.OBJ_THRESHOLD
    n = torch.stack([(ok & (cl == c)).sum(1) for c in range(3)], 1).cpu().numpy()
    return n


for (B, S, stage) in ((8, 576, 1), (8, 576, 2), (4, 832, 1)):
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=stage, seed=0)
    net.set_batch(synthetic_batch(B, S, seed=1234))
    torch.manual_seed(1234)
    net.shuffle_seed = 1234
    for step in range(0, 121):
        net.train_step(None, want_loss=False)
        if step in (0, 5, 20, 60, 120):
            torch.cuda.synchronize()
            n = lists(net, B, S)
            print("train B=%d %d^2 stage %d, step %3d: per (image, class) list lengths max %d, mean %.0f, over 8192: %d of %d" % (B, S, stage, step, n.max(), n.mean(), (n > 8192).sum(), n.size), flush=True)
    del net
B, S = 32, 576
net = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
batch = synthetic_batch(B, S, seed=1234)
net._set_inputs(batch["images"], batch["clip_window"])
net.build_infer_program(graph=False)
net.infer()
torch.cuda.synchronize()
n = lists(net, B, S)
print("infer B=%d %d^2 (random init): per (image, class) list lengths max %d, mean %.0f, over 8192: %d of %d; detections %d" % (B, S, n.max(), n.mean(), (n > 8192).sum(), n.size, int(net.det_count.sum())), flush=True)
