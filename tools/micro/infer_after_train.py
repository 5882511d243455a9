import sys, time
sys.path.insert(0, ".")
import torch, bench
from bench import YOLONet, synthetic_batch, repeated
import numpy as np
dev = torch.device("cuda:0")
S = 576
def infer_rate(tag):
    B = 32
    net = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
    batch = synthetic_batch(B, S, seed=1234)
    net._set_inputs(batch["images"], batch["clip_window"])
    if '--noautotune' not in sys.argv: net.autotune()
    net.build_infer_program(graph=True)
    for _ in range(3): net.infer()
    med, ts = repeated(lambda: net.infer(), 10, 5, 1, dev)
    print(tag, "infer img/s %.0f" % (B * 10 / med), flush=True)
    del net; torch.cuda.empty_cache()
infer_rate("fresh")
infer_rate("fresh again")
import disyolo_amd.lib as L
print("tuned entries", len(L.TUNED), flush=True)
for stage in (1, 2):
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=8, stage=stage, seed=0)
    net.set_batch(synthetic_batch(8, S, seed=4321)); net.shuffle_seed = 99
    net.autotune(); net.build_program()
    for _ in range(3): net.train_step(None, want_loss=False)
    med, ts = repeated(lambda: net.train_step(None, want_loss=False), 10, 5, 1, dev)
    print("stage", stage, "train img/s %.0f" % (8 * 10 / med), flush=True)
    if stage == 2:
        del net; torch.cuda.empty_cache()
    infer_rate("after stage %d" % stage)
    saved = dict(L.TUNED); L.TUNED.clear()
    infer_rate("after stage %d, tile table cleared first" % stage)
    L.TUNED.update(saved)
