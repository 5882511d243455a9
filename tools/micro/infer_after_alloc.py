"""Does inference slow down after the process has allocated and freed a lot of device memory (no compute)?"""
import sys
sys.path.insert(0, ".")
import torch, bench
from bench import YOLONet, synthetic_batch, repeated
dev = torch.device("cuda:0")
S = 576
def infer_rate(tag):
    B = 32
    net = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
    batch = synthetic_batch(B, S, seed=1234)
    net._set_inputs(batch["images"], batch["clip_window"])
    net.build_infer_program(graph=True)
    for _ in range(3): net.infer()
    med, ts = repeated(lambda: net.infer(), 10, 5, 1, dev)
    print(tag, "infer img/s %.0f" % (B * 10 / med), flush=True)
    del net; torch.cuda.empty_cache()
infer_rate("fresh")
# many tensors of training-net-like sizes, touched, then all freed
ts = [torch.zeros(n, dtype=torch.bfloat16, device=dev) for n in [85 * 2**20, 42 * 2**20, 21 * 2**20, 10 * 2**20, 5 * 2**20] * 40]
torch.cuda.synchronize()
print("allocated GB", sum(t.numel() for t in ts) * 2 / 1e9, flush=True)
infer_rate("while 13 GB of other tensors are alive")
del ts; torch.cuda.empty_cache()
infer_rate("after they were freed")
infer_rate("again")
