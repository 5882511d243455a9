"""Host time to enqueue one recorded training step vs the GPU time of the step (GPU box): is the step
host-bound?   usage: python tools/host_enqueue.py [stage] [plain]   (stage 1: the pipelined step unless "plain")"""
import sys, time
sys.path.insert(0, ".")
import torch
import bench  # noqa: F401  (sets up the package alias)
from bench import YOLONet, synthetic_batch
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
B, S = 8, 576
net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=stage, seed=0)
net.set_batch(synthetic_batch(B, S, seed=1234))
net.shuffle_seed = 1234
import os
cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tune_train_B8_576_stage%d.json" % stage)
if os.path.exists(cache):
    net.autotune(cache=cache)
pipe = stage == 1 and "plain" not in sys.argv
net.build_program(pipeline_backbone=pipe, overlap_tail=not pipe)
if pipe:
    net.prime_pipeline()
for _ in range(5):
    net.train_step(None, want_loss=False)
torch.cuda.synchronize()
for n in (1, 10, 30):
    t0 = time.perf_counter()
    for _ in range(n):
        net.train_step(None, want_loss=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("stage %d, %2d steps: host enqueue %.3f ms/step, wall (to sync) %.3f ms/step, %d commands"
          % (stage, n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, net._prog.size()))
