"""Per-layer comparison of the HIP forward against the oracle (debug aid, GPU box only)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch, numpy as np
import disyolo_amd
import disyolo_oracle as O
from test_gpu_net import make_net, oracle_params, rel_err

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
training = (sys.argv[2] == "train") if len(sys.argv) > 2 else True
B, S = 2, int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
net = make_net(dev, training, stage, B=B, S=S, seed=1)
b = O.synthetic_batch(B, S, seed=11)
p0 = oracle_params(net)
lock = O.default_lock(stage)
if training:
    net.set_batch(b)
    net._forward_layers(True)
else:
    net.forward(b["images"], b["clip_window"], [0.1], False)
torch.cuda.synchronize()
taps = {}
O.build_network(p0, b["images"], training, lock, {}, taps, quant=O.bf16_ste)
taps32 = {}
O.build_network(p0, b["images"], training, lock, {}, taps32)
for l in net.layers:
    if l.idx < 50 and l.idx % 10: continue
    r, amax, wmax = rel_err(l.act, taps["act%d" % l.idx])
    r32, _, _ = rel_err(taps["act%d" % l.idx], taps32["act%d" % l.idx])
    flag = "  <<<<" if r > 2e-2 else ""
    print("layer %2d %-3s lock=%d  hip-vs-oracle(bf16) %.3e  | oracle(bf16)-vs-oracle(f32) %.3e%s" % (l.idx, l.kind, l.lock, r, r32, flag))
