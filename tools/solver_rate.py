"""The loop a user runs, end to end on the GPU: polygon records -> defect_train (the GPU data pipeline: rasterised masks, scale / crop /
flip, blur / noise / light) -> Solver.train (recorded step) at B = 8, 576^2, stage 1 -- images per second over the last STEPS - 50 steps
(host clock at every data.get(), device synchronised at the end).

    python tools/solver_rate.py [--steps 300] [--pipeline auto|off]        (DISYOLO_FEED_STREAM=0: the feed on the caller's stream)"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import disyolo_amd  # noqa: E402,F401
from disyolo_amd import config as cfg  # noqa: E402
from disyolo_amd.net import YOLONet  # noqa: E402
from disyolo_amd.solver import Solver  # noqa: E402
from disyolo_amd.train_data import defect_train  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from train_synthetic import synthetic_labels  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--pipeline", default="auto")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=576)
ap.add_argument("--stage", type=int, default=1)
ap.add_argument("--json", action="store_true", help="one JSON line instead of the sentence (bench.py's secondary)")
args = ap.parse_args()
dev = torch.device("cuda:0")
labels = synthetic_labels(np.random.RandomState(0), 64)
data = defect_train(labels, batch_size=args.batch, image_size=args.size, device=dev, rng=np.random.RandomState(1))
net = YOLONet(training=True, device=dev, image_size=args.size, batch_size=args.batch, stage=args.stage, seed=0)
cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tune_train_B8_576_stage%d.json" % args.stage)
stamps = []
inner = data.get


def get():
    stamps.append(time.perf_counter())
    return inner()


data.get = get
if hasattr(data, "get_device") and os.environ.get("SOLVER_RATE_HOST_LABELS") != "1":
    inner_dev = data.get_device

    def get_device():
        stamps.append(time.perf_counter())
        return inner_dev()
    data.get_device = get_device
    inner = inner_dev
else:
    data.get_device = None
# the data pipeline alone first (its own GPU time per batch)
for _ in range(5):
    inner()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    inner()
torch.cuda.synchronize()
t_data = (time.perf_counter() - t0) / 30
if args.batch == 8 and args.size == 576 and os.path.exists(cache):
    net.autotune(cache=cache)       # (the committed tile table: loaded, nothing is timed)
with tempfile.TemporaryDirectory() as out:
    solver = Solver(net, data, output_dir=out, max_iter=args.steps, summary_iter=10 ** 9, save_iter=10 ** 9, log=lambda *_: None,
                    pipeline_backbone=None if args.pipeline == "auto" else False)
    hist = solver.train()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
skip = 50
n = len(stamps) - skip
dt = (t_end - stamps[skip]) / n
if args.json:
    import json
    print(json.dumps({"workload": "Solver.train over train_data.defect_train (GPU data pipeline), B%d %dx%d, stage %d" % (args.batch, args.size, args.size, args.stage),
                      "value": round(args.batch / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3), "steps": n,
                      "data_pipeline_alone_ms_per_batch": round(t_data * 1e3, 3), "finite_losses": int(np.isfinite(hist).sum()),
                      "records": len(labels)}))
    sys.exit(0)
print("data pipeline alone %.3f ms per batch; Solver.train (stage %d): %d steps, %.3f ms per step = %.1f images/s (pipeline %s, feed stream %s); "
      "finite losses %d of %d" % (t_data * 1e3, args.stage, n, dt * 1e3, args.batch / dt, args.pipeline, os.environ.get("DISYOLO_FEED_STREAM", "1"),
                                  int(np.isfinite(hist).sum()), len(hist)))
