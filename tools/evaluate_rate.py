"""evaluate() (the mirror of calculate_test_map.py:180-347: one image at a time through an inference net of batch size 1, boxes
un-letterboxed, masks cropped / resized / pasted, mAP + mIoU) at 576^2 over N synthetic images -- seconds per image, split like the
reference's own timing (prediction / crop + assemble).   python tools/evaluate_rate.py [N]      (DISYOLO_EVAL_REPLAY=0: eager launches)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disyolo_amd  # noqa: E402,F401
from disyolo_amd import evaluate as E  # noqa: E402
from disyolo_amd.net import YOLONet  # noqa: E402

as_json = "--json" in sys.argv
args_ = [a for a in sys.argv[1:] if a != "--json"]
N = int(args_[0]) if args_ else 48
S = 576
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
images, recs, sizes, merged, index = {}, {}, {}, {}, []
for k in range(N):
    h, w = int(rng.randint(400, 900)), int(rng.randint(400, 900))
    name = "img%03d" % k
    index.append(name)
    images[name] = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    sizes[name] = [h, w]
    yy, xx = np.mgrid[0:h, 0:w]
    objs, mm = [], np.zeros((h, w), np.uint8)
    for j in range(3):
        cy, cx, ry, rx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w, rng.uniform(0.1, 0.3) * h, rng.uniform(0.1, 0.3) * w
        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        c = int(rng.randint(0, 3))
        objs.append({"imageid": name, "classid": c, "difficult": 0, "mask": m})
        mm[m] = c + 1
    recs[name], merged[name] = objs, mm
emap = E.MAP(recs, sizes, index, merged, net_size=S)
net = YOLONet(training=False, device=dev, image_size=S, batch_size=1, stage=1, seed=0)
# heads with enough spread that detections exist (random initialisation gives none above the threshold)
g = torch.Generator().manual_seed(5)
for i in (59, 67, 75):
    l = net.by_idx[i]
    l.w.copy_(torch.randn(l.w.shape, generator=g).to(dev) * 0.05)
net.refresh_weights()
E.evaluate(net, {k: images[k] for k in index[:4]}, E.MAP({k: recs[k] for k in index[:4]}, {k: sizes[k] for k in index[:4]}, index[:4],
                                                            {k: merged[k] for k in index[:4]}, net_size=S), det_thresh=0.05)     # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
thresh_out, acc, timing = E.evaluate(net, images, emap, det_thresh=0.05)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if as_json:
    import json
    print(json.dumps({"workload": "evaluate() (calculate_test_map.py loop): batch-1 inference at 576x576 + paste + mask mAP / mIoU over %d images of 400-900 px" % N,
                      "value": round(N / dt, 1), "unit": "images/sec", "ms_per_image": round(dt / N * 1e3, 3),
                      "prediction_ms_per_image": round(timing["prediction_s"] / N * 1e3, 3),
                      "crop_assemble_ms_per_image": round(timing["crop_assemble_s"] / N * 1e3, 3)}))
    sys.exit(0)
print("evaluate(): %d images of 400-900 px, %.2f ms per image wall (prediction %.2f ms, crop + assemble %.2f ms), replay %s, mAP rows %d"
      % (N, dt / N * 1e3, timing["prediction_s"] / N * 1e3, timing["crop_assemble_s"] / N * 1e3,
         os.environ.get("DISYOLO_EVAL_REPLAY", "1"), len(thresh_out)))
