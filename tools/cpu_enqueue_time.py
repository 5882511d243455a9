"""How long does the host take to enqueue one recorded step (vs the GPU's time to run it)?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch
dev = torch.device("cuda:0")
net = YOLONet(training=True, device=dev, image_size=576, batch_size=8, stage=1, seed=0)
net.set_batch(synthetic_batch(8, 576, seed=1234))
net.shuffle_seed = 1
net.autotune()
net.build_program()
for _ in range(5):
    net.train_step(None, want_loss=False)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    net.train_step(None, want_loss=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, GPU drain after the last enqueue %.3f ms, total %.3f ms/step"
      % ((t1 - t0) / N * 1e3, (t2 - t1) * 1e3, (t2 - t0) / N * 1e3))
