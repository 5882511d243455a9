"""Host time to enqueue one recorded training step vs the GPU time of the step (GPU box): is the step
host-bound?   usage: python tools/host_enqueue.py [stage]"""
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
net.build_program()
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
