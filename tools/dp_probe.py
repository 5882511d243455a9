"""in-list gradient exchange on one rank: where the time goes (host enqueue vs GPU), with and without the collectives.
python tools/dp_probe.py   (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import disyolo_amd  # noqa: E402,F401
from disyolo_amd import lib as L  # noqa: E402
from disyolo_amd.net import YOLONet  # noqa: E402
from disyolo_amd.synth import synthetic_batch  # noqa: E402
from disyolo_amd.dp import enable_data_parallel  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


def run(tag, dp, overlap, inlist=True):
    net = YOLONet(training=True, device=dev, image_size=576, batch_size=8, stage=1, seed=0)
    if dp:
        enable_data_parallel(net, inlist=inlist)
    net.set_batch(synthetic_batch(8, 576, seed=1234))
    net.shuffle_seed = 1234
    net.build_program(overlap_tail=overlap)
    for _ in range(5):
        net.train_step(None, want_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(20):
        h0 = time.perf_counter()
        net.train_step(None, want_loss=False)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-34s %.3f ms/step   host enqueue %.3f ms/step" % (tag, dt / 20 * 1e3, host / 20 * 1e3), flush=True)
    del net
    torch.cuda.empty_cache()


# a bare collective: host time of the call and GPU time
comm = L.Comm(0, 1, lambda ident: ident)
buf = torch.zeros(10 << 20, device=dev)
side = torch.cuda.Stream()
for st, name in ((torch.cuda.current_stream(), "current stream"), (side, "a side stream")):
    with torch.cuda.stream(st):
        comm.allreduce(buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            comm.allreduce(buf)
        h = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("bare 40 MB all-reduce on %s: host %.1f us per call, with sync %.1f us" % (name, h / 20 * 1e6, (time.perf_counter() - t0) / 20 * 1e6), flush=True)

run("plain, overlap", False, True)
run("dp in-list, overlap", True, True)
run("dp in-list, joined", True, False)
os.environ["DISYOLO_DP_DRY"] = "1"
run("dp in-list DRY (no collectives), overlap", True, True)
os.environ.pop("DISYOLO_DP_DRY")
run("dp cut list", True, False, inlist=False)
torch.cuda.synchronize()
dist.destroy_process_group()
