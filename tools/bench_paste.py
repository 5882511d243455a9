"""Time disyolo_mask_paste on one evaluate()-sized image: 30 detections, 288x288 masks, 754x1008 image."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import disyolo_amd
from disyolo_amd import postprocess as P
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
n, size, net, h, w = 30, 288, 576, 754, 1008
masks = torch.rand(n, size, size, device=dev)
box = np.zeros((n, 6), np.float32)
for k in range(n):
    y1, x1 = rng.uniform(0.15, 0.6), rng.uniform(0.0, 0.6)
    box[k, :4] = [y1, x1, y1 + rng.uniform(0.05, 0.3), x1 + rng.uniform(0.05, 0.4)]
    box[k, 4] = rng.randint(0, 3)
for want_full in (True, False):
    for _ in range(3):
        P.paste_detections(box, masks, h, w, net, want_full)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        P.paste_detections(box, masks, h, w, net, want_full)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    out_b = (n if want_full else 0) * h * w + h * w
    print("full masks %s: %.3f ms per image end to end (host rect math + H2D + kernel), %.1f MB written" % (want_full, ms, out_b / 1e6))
