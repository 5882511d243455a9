"""End-to-end sanity of the recorded training step: overfit ONE synthetic batch for N steps (stage 1 and 2) and
print the loss curve; the recorded step (optimizer sweeps overlapped with backward) must make it fall.
usage: python tools/overfit_check.py [steps]"""
import sys
sys.path.insert(0, ".")
import torch, bench
from bench import YOLONet, synthetic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
for stage in (1, 2):
    net = YOLONet(training=True, device=dev, image_size=192, batch_size=4, stage=stage, seed=0)
    net.set_batch(synthetic_batch(4, 192, seed=7))
    net.shuffle_seed = 11
    net.build_program()
    curve = []
    for t in range(n):
        loss = net.train_step(None)
        if t % (n // 10) == 0 or t == n - 1:
            curve.append(round(float(loss.cpu()), 2))
    print("stage", stage, "total loss every %d steps:" % (n // 10), curve, flush=True)
    assert all(c == c for c in curve), "NaN"
    assert curve[-1] < 0.6 * curve[0], "loss did not fall"
