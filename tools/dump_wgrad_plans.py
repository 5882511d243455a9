"""weight-gradient plan per trainable layer (GPU box): python tools/dump_wgrad_plans.py [stage]"""
import sys
sys.path.insert(0, ".")
import bench, torch
from bench import YOLONet
from disyolo_amd import lib as L
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = YOLONet(training=True, device=torch.device("cuda:0"), image_size=576, batch_size=8, stage=stage, seed=0)
for l in n.layers:
    if l.lock or l.wgrad_desc is None:
        continue
    kind, tn, ring, splits = L.conv2d_wgrad_plan(l.wgrad_desc)
    M = 8 * l.Ho * l.Wo
    cin = l.cin
    gf = 2 * M * cin * l.cout * l.k * l.k / 1e9
    print(l.idx, l.kind, "k%d s%d" % (l.k, l.stride), "%4d->%4d" % (cin, l.cout), "M=%6d" % M, "GF=%6.2f" % gf,
          "tapfused" if kind == 1 else "im2col  ", "tile_n", tn, "ring", ring, "splits", splits,
          "slabMB=%.1f" % (splits * l.k * l.k * cin * l.cout * 4 / 1e6 if splits > 1 else 0), "concat" if l.src_up else "")
