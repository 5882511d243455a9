"""Data-parallel gradient exchange on CPU: two gloo ranks, the real bucket planner and
GradientAllReduce driven through the same begin/on_layer_done/finish protocol the train
step uses.  Checks: every arena element is summed exactly once, buckets fire in backward
order, and the result scaled by 1/world equals the mean of the per-rank gradients."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from disyolo_amd.dp import GradientAllReduce
from disyolo_amd.net import build_topology


class StubNet:
    """CPU stand-in with the attributes GradientAllReduce reads from YOLONet."""

    def __init__(self, rank):
        self.layers = build_topology(3, 3)
        off = 0
        self.arena_slices = {}
        for l in self.layers:
            l.lock = l.idx <= 52
        for l in self.layers:
            if l.lock:
                continue
            n = l.k * l.k * l.cin * l.cout // 64 + 1        # shrunk: keep the test light
            self.arena_slices["yolo/convolutional%d/weights" % l.idx] = (off, n)
            off += n
            if l.kind == "lin":
                self.arena_slices["yolo/convolutional%d/biases" % l.idx] = (off, l.cout)
                off += l.cout
        self.n_decay = off
        for l in self.layers:
            if not l.lock and l.kind != "lin":
                for leaf in ("gamma", "beta"):
                    self.arena_slices["yolo/convolutional%d/BatchNorm/%s" % (l.idx, leaf)] = (off, l.cout)
                    off += l.cout
        self.n_params = off
        g = torch.Generator().manual_seed(100 + rank)
        self.grad_arena = torch.randn(off, generator=g)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = StubNet(rank)
    local = net.grad_arena.clone()
    dp = GradientAllReduce(net, bucket_mb=0.02)
    fired = []
    dp.begin_step()
    for l in reversed(net.layers):
        if l.lock:
            continue
        before = len(dp.works)
        dp.on_layer_done(l)
        if len(dp.works) > before:
            fired.append(l.idx)
    dp.finish()
    others = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(others, local)
    want = torch.stack(others).sum(0)
    ok = torch.allclose(net.grad_arena, want, rtol=0, atol=1e-6)
    covered = sorted((o, o + c) for _, o, c in dp.buckets) + [(net.n_decay, net.n_params)]
    contiguous = covered[0][0] == 0 and all(a[1] == b[0] for a, b in zip(covered, covered[1:])) and covered[-1][1] == net.n_params
    if rank == 0:
        torch.save({"ok": bool(ok), "fired": fired, "nb": len(dp.buckets), "contiguous": contiguous,
                    "mean_ok": bool(torch.allclose(net.grad_arena / world, torch.stack(others).mean(0), atol=1e-6))}, out)
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["ok"] and r["mean_ok"] and r["contiguous"]
    assert r["nb"] >= 3
    assert r["fired"] == sorted(r["fired"], reverse=True) and len(r["fired"]) == r["nb"]
