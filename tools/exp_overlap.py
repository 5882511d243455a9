"""Experiment (GPU box): how much does a concurrent frozen-backbone forward (layers 1-52 of a
second net, on its own stream) slow the full training step?  Estimates the gain of overlapping
step t+1's backbone with step t's backward (cross-step software pipeline, stage 1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import disyolo_amd
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

dev = torch.device("cuda:0")
B, S = 8, 576
net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
net.set_batch(synthetic_batch(B, S, seed=1))
net.build_program()
bb = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
bb._set_inputs(synthetic_batch(B, S, seed=2)["images"], [[0, 0, 1, 1]] * B)
prog = L.CmdList()
with prog:
    for l in bb.layers[:52]:
        bb._forward_layer(l, False)
side = torch.cuda.Stream(device=dev)

def run(n, with_bb, delay_bb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net.train_step(None, want_loss=False)
        if with_bb:
            with torch.cuda.stream(side):
                prog.run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(5):
    net.train_step(None, want_loss=False)
print("step alone            : %.3f ms" % run(20, False, False))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    prog.run()
torch.cuda.synchronize()
print("backbone alone        : %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
print("step + concurrent bb  : %.3f ms" % run(20, True, False))
