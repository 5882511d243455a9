import sys
sys.path.insert(0, ".")
import torch, bench
from bench import YOLONet, synthetic_batch
dev = torch.device("cuda:0")
for stage in (1, 2):
    net = YOLONet(training=True, device=dev, image_size=576, batch_size=8, stage=stage, seed=0)
    net.set_batch(synthetic_batch(8, 576, seed=1)); net.shuffle_seed = 1
    net.autotune(cache="gpurun_out/ab_tune.json" if stage == 1 else None)
    net.compute_losses(); net.backward()
    torch.cuda.synchronize()
    fused = [l.idx for l in net.layers if l.bwd_part_rows]
    bn = [l.idx for l in net.layers if not l.lock and l.kind != "lin"]
    print("stage", stage, "fused", len(fused), "of", len(bn), fused)
