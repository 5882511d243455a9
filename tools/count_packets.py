"""queue packets per lane of the recorded training step (launches, event records, event waits): python tools/count_packets.py [stage]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd  # noqa: F401
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
for overlap in (False, True):
    net = YOLONet(training=True, device=dev, image_size=576, batch_size=8, stage=stage, seed=0)
    net.set_batch(synthetic_batch(8, 576, seed=1))
    net.shuffle_seed = 1
    net.build_program(overlap_tail=overlap)
    p = net._prog
    print("stage %d overlap_tail=%s: %d commands" % (stage, overlap, p.size()))
    for lane in (0, 1):
        print("   lane %d: %3d launches, %2d event records, %2d event waits" % (lane, p.count("launches", lane), p.count("records", lane), p.count("waits", lane)))
    del net
    torch.cuda.empty_cache()
