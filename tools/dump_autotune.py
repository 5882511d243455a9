"""Dump the in-sequence autotuner's timing table (GPU box): python tools/dump_autotune.py [stage] > profiles/..."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
net = YOLONet(training=True, device=dev, image_size=576, batch_size=8, stage=stage, seed=0)
net.set_batch(synthetic_batch(8, 576, seed=1234))
tuner_holder = {}
orig = L.ConvTuner.commit
def commit(self, min_gain=0.02):
    tuner_holder["t"] = self.table()
    return orig(self, min_gain)
L.ConvTuner.commit = commit
picks = net.autotune(reps=5)
tab = tuner_holder["t"]
cands = (0,) + L.TUNE_CANDIDATES
print("# in-sequence conv tile timings, us (median of 5 eager passes), B=8 576^2 stage %d; key = (B,H,W,C0,C1,Ho,Wo,Cout,k,stride,in_div)" % stage)
print("%-52s" % "shape" + " ".join("%7s" % ("heur" if c == 0 else "t%x" % c) for c in cands) + "   pick")
for key, row in tab.items():
    print("%-52s" % str(key) + " ".join("%7.1f" % (row[c] * 1e3) if c in row else "      -" for c in cands)
          + "   " + ("t%x" % picks[key] if picks[key] else "heur"))
