"""Experiment (GPU box): locked-backbone forward for 8 images as ONE chain vs TWO concurrent
chains of 4 images on two streams (the images are independent through locked layers)."""
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
S = 576
def backbone_prog(B, seed):
    net = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
    net._set_inputs(synthetic_batch(B, S, seed=seed)["images"], [[0, 0, 1, 1]] * B)
    prog = L.CmdList()
    with prog:
        for l in net.layers[:52]:
            net._forward_layer(l, False)
    return net, prog
n8, p8 = backbone_prog(8, 1)
n4a, p4a = backbone_prog(4, 2)
n4b, p4b = backbone_prog(4, 3)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def two():
    with torch.cuda.stream(s1): p4a.run()
    with torch.cuda.stream(s2): p4b.run()
def seq():
    p4a.run(); p4b.run()
print("B=8 one chain        : %.3f ms" % t(lambda: p8.run()))
print("2 x B=4 sequential   : %.3f ms" % t(seq))
print("2 x B=4 two streams  : %.3f ms" % t(two))
