"""PCIe-inclusive training rate (GPU box): every step's batch comes from pinned HOST memory -- images f32, masks
uint8, targets, boxes: 90 MB per step at B=8, 576x576 -- through a copy stream into a device staging set (double
buffered), then device-to-device into the network's input buffers at the start of the step.  Prints the rate with
resident inputs (what bench.py reports), with a synchronous feed, and with the overlapped feed.
usage: python tools/feed_rate.py [stage] [plain]"""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from bench import YOLONet, synthetic_batch, repeated
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
B, S = 8, 576
net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=stage, seed=0)
batch = synthetic_batch(B, S, seed=1234)
net.set_batch(batch)
net.shuffle_seed = 1234
import os
cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tune_train_B8_576_stage%d.json" % stage)
net.autotune(cache=cache if os.path.exists(cache) else None)
pipe = stage == 1 and "plain" not in sys.argv       # (round 6: the step bench.py and Solver run in stage 1 is the pipelined one)
net.build_program(pipeline_backbone=pipe, overlap_tail=not pipe)
if pipe:
    net.prime_pipeline()
keys = ("images", "clip_window", "true_boxes", "true_masks", "yolo1", "yolo2", "yolo3")
host = {k: torch.as_tensor(batch[k]).contiguous() for k in keys}
host["true_masks"] = host["true_masks"].to(torch.uint8)
host = {k: v.pin_memory() for k, v in host.items()}
nbytes = sum(v.numel() * v.element_size() for v in host.values())
from disyolo_amd.feed import HostFeeder
feeder = HostFeeder(net)

def resident():
    net.train_step(None, want_loss=False)

def sync_feed():
    net.set_batch({k: v.to(dev, non_blocking=False) for k, v in host.items()})
    net.train_step(None, want_loss=False)

def overlapped():
    feeder.submit(host)                                # next batch: flies during this step
    feeder.step(want_loss=False)

for _ in range(5):
    resident()
med, _ = repeated(resident, 20, 5, 1, dev)
print("resident inputs      : %.0f img/s  (%.3f ms/step)" % (B * 20 / med, med / 20 * 1e3))
for _ in range(3):
    sync_feed()
med, _ = repeated(sync_feed, 20, 5, 1, dev)
print("synchronous host feed: %.0f img/s  (%.3f ms/step), %.1f MB per step" % (B * 20 / med, med / 20 * 1e3, nbytes / 1e6))
feeder.submit(host)
for _ in range(3):
    overlapped()
med, _ = repeated(overlapped, 20, 5, 1, dev)
print("overlapped host feed : %.0f img/s  (%.3f ms/step)" % (B * 20 / med, med / 20 * 1e3))
