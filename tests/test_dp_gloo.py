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


class _FakeList:
    """stands in for the recorded command list (csrc/runtime.hip) on the CPU: run(a, b) "executes" the commands a..b-1 --
    here: the producers registered at those positions fill their tensors -- and logs the call"""

    def __init__(self):
        self.n = 0
        self.producers = {}      # command index -> callable
        self.log = []

    def size(self):
        return self.n

    def add(self, fn=None):
        if fn is not None:
            self.producers[self.n] = fn
        self.n += 1

    def run(self, a=0, b=None, fork=True, join=True):
        b = self.n if b is None else b
        self.log.append(("run", a, b))
        for i in range(a, b):
            if i in self.producers:
                self.producers[i]()


def _syncbn_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    torch.set_num_threads(2)
    net = YOLONet(training=True, stage=1, seed=3, plan_only=True)
    dp = enable_data_parallel(net, bucket_mb=4.0)
    net.sync_bn = True                 # (enable_sync_bn allocates device buffers; the cut loop only needs the flag's effect: marks)
    net.use_side_lane = True
    prog = _FakeList()
    marks = []
    net._rec = (prog, marks)
    import disyolo_amd.lib as L
    bn = [l for l in net.layers if not l.lock and l.kind != "lin"]
    sums = {}
    log = prog.log
    # ---- "record" a step the way _record_step does: forward (a SyncBN cut after every trainable layer's partial sums, on the
    # lane that produced them: the three head branches run on the side lane), then backward in backward_order() (a SyncBN
    # cut per batch-norm backward, a bucket mark where a bucket's last member is done)
    for l in bn:
        t = torch.zeros(l.cout, 2, dtype=torch.float64)
        sums[("f", l.idx)] = t
        prog.add()                                                                  # the conv
        prog.add(lambda t=t, l=l: t.fill_(float(rank + 1) * l.idx))                 # bn_partial_sums
        L.CURRENT_LANE = 1 if l.idx in net.HEAD_LAYERS else 0
        net._sync_sums(t)
        prog.add()                                                                  # bn_finalize_sums + activation
    L.CURRENT_LANE = 0
    dp.begin_step()
    g = torch.Generator().manual_seed(500 + rank)
    local = torch.randn(net.n_params, generator=g)
    for l in net.backward_order():
        if l.lock:
            continue
        if l.kind != "lin":
            t = torch.zeros(l.cout, 2, dtype=torch.float64)
            sums[("b", l.idx)] = t
            prog.add(lambda t=t, l=l: t.fill_(float(rank + 1) * 0.5 * l.idx))       # bn_bwd_reduce
            net._sync_sums(t)
            prog.add()                                                              # bn_bwd_apply_sums
        prog.add()                                                                  # weight gradient
        bi = dp.completes_bucket(l)
        if bi is not None:
            marks.append((prog.size(), bi))
    net._rec = None
    bwd_end = prog.size()
    prog.add(lambda: log.append(("adam",)))                                         # optimizer
    # the gradients exist when the list is replayed, not when it is recorded
    prog.producers[0] = lambda: net.grad_arena.copy_(local)
    net._prog, net._prog_marks, net._bwd_end, net._progs, net._graph = prog, marks, bwd_end, None, None
    fired = []
    orig_fire = dp.fire
    dp.fire = lambda bi: (fired.append((prog.log[-1][2], bi)), orig_fire(bi))
    net.run_program()
    others = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(others, local)
    want = torch.stack(others).sum(0)
    ok_grad = float((net.grad_arena - want).abs().max() / want.abs().max()) < 1e-6
    tot = sum(range(1, world + 1))
    ok_sums = all(bool((t == (tot * (l if k == "f" else 0.5 * l))).all()) for (k, l), t in sums.items())
    runs = [e for e in log if e[0] == "run"]
    cuts = [idx for idx, _ in marks]
    # segments: consecutive, cut exactly at the marks, the backward tail joined before finish(), Adam after it
    seg_ok = [r[1:] for r in runs] == list(zip([0] + cuts, cuts + [bwd_end])) + [(bwd_end, prog.size())]
    fire_ok = fired == [(idx, what) for idx, what in marks if not isinstance(what, tuple)]
    if rank == 0:
        torch.save({"ok_grad": ok_grad, "ok_sums": ok_sums, "seg_ok": seg_ok, "fire_ok": fire_ok, "n_sync": sum(isinstance(w, tuple) for _, w in marks),
                    "n_bucket": sum(not isinstance(w, tuple) for _, w in marks), "n_bn": len(bn), "adam_last": log[-1] == ("adam",),
                    "side_lane_syncs": sum(1 for _, w in marks if isinstance(w, tuple) and w[2] == 1)}, out)
    dist.destroy_process_group()


def test_syncbn_cut_list_through_run_program_two_ranks(tmp_path):
    """SyncBN under data parallelism (SURVEY.md 8e option): the recorded step is cut where a layer's per-channel sums exist
    and where a gradient bucket becomes final; YOLONet.run_program replays the segments in order and issues the collectives
    between them.  Two gloo ranks, the real run_program / _sync_sums / bucket objects over a stand-in command list whose
    "kernels" produce rank-dependent sums: every cut tensor ends up as the sum over the ranks, every bucket fires right after
    the segment that completes it, the optimizer runs after finish(), and the head branches' sums are cut on the side lane."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "s.pt")
    mp.spawn(_syncbn_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["ok_grad"] and r["ok_sums"] and r["seg_ok"] and r["fire_ok"] and r["adam_last"], r
    assert r["n_bn"] == 26 and r["n_sync"] == 2 * 26 and r["n_bucket"] >= 5 and r["side_lane_syncs"] == 3
