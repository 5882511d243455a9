"""Data-parallel gradient exchange on CPU: two gloo ranks drive the REAL objects of the train step --
a plan-only ``YOLONet`` (the reference's variables in the flat arena, stage 1 and stage 2), its
``backward_order()``, the bucket planner and ``GradientAllReduce`` -- through the same
begin_step / on_layer_done / finish protocol ``YOLONet.train_step`` uses.  Checks: every arena
element is summed exactly once, a bucket fires when its last member layer is done (the heads-first
visiting order is not descending), the result scaled by 1/world equals the mean of the per-rank
gradients, rank 0's variables are broadcast, and the bf16 wire format stays within its rounding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from disyolo_amd.dp import enable_data_parallel
from disyolo_amd.net import YOLONet


def _worker(rank, world, port, out, stage, wire):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # loopback: no resolution of the container's hostname
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    torch.set_num_threads(2)
    net = YOLONet(training=True, stage=stage, seed=10 + rank, plan_only=True)      # different init per rank
    w_before = net.arena.clone()
    dp = enable_data_parallel(net, bucket_mb=4.0, wire=wire)
    # broadcast: every rank now holds rank 0's variables
    ref = [torch.zeros_like(net.arena) for _ in range(world)]
    dist.all_gather(ref, net.arena)
    same_after_bcast = all(torch.equal(r, ref[0]) for r in ref)
    differed_before = bool((w_before != net.arena).any()) if rank != 0 else True
    g = torch.Generator().manual_seed(100 + rank)
    net.grad_arena.copy_(torch.randn(net.n_params, generator=g))
    local = net.grad_arena.clone()
    fired = []
    dp.begin_step()
    for l in net.backward_order():
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
    err = float((net.grad_arena - want).abs().max() / want.abs().max())
    covered = sorted((o, o + c) for _, o, c in dp.buckets) + [(net.n_decay, net.n_params)]
    contiguous = covered[0][0] == 0 and all(a[1] == b[0] for a, b in zip(covered, covered[1:])) and covered[-1][1] == net.n_params
    # a bucket fires exactly when its last member in visiting order is done
    order = [l.idx for l in net.backward_order() if not l.lock]
    expect_fired = []
    for bi, mem in enumerate(dp.members):
        expect_fired.append(max(mem, key=order.index))
    expect_fired.sort(key=order.index)
    if rank == 0:
        torch.save({"err": err, "fired": fired, "expect_fired": expect_fired, "nb": len(dp.buckets), "contiguous": contiguous,
                    "bcast": same_after_bcast, "n_params": net.n_params}, out)
    else:
        torch.save({"differed_before": differed_before}, out + ".r1")
    dist.destroy_process_group()


@pytest.mark.parametrize("stage,wire", [(1, "f32"), (1, "bf16"), (2, "f32")])
def test_bucketed_allreduce_two_ranks_through_the_real_protocol(tmp_path, stage, wire):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker, args=(2, port, out, stage, wire), nprocs=2, join=True)
    r = torch.load(out)
    assert r["contiguous"] and r["bcast"] and torch.load(out + ".r1")["differed_before"]
    assert r["n_params"] == (21_070_737 if stage == 1 else 61_655_665)
    assert r["nb"] >= 5
    assert r["fired"] == r["expect_fired"] and len(r["fired"]) == r["nb"]
    # f32 wire: exact up to the summation order of two ranks; bf16 wire: each contribution rounded to 8 bits
    assert r["err"] < (1e-6 if wire == "f32" else 2 ** -7)
