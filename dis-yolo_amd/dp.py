"""Data-parallel training: gradient all-reduce over RCCL (xGMI), overlapped with backward.

The reference is single-GPU (yolo/config.py:18); every loss term is a mean over the batch
of per-image sums (yolo/yolo3_net_pos.py:692-726,858), so with equal local batches the
global gradient is the mean of the per-rank gradients -- one exchange step per iteration
and nothing else (batch-norm statistics stay local).

Gradients live in one flat f32 arena ordered by layer index.  Backward visits layers in
descending order, so a bucket is a contiguous arena slice [layer a .. layer b] that becomes
final when layer a's weight gradient has been enqueued.  Each bucket is all-reduced
asynchronously (RCCL's own stream, ordered after the producing kernels) while the backward
pass of the earlier layers keeps the compute stream busy; the optimizer waits on all of
them and applies the 1/world_size scale inside the Adam kernel.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .lib import DisyoloError


def plan_buckets(layer_spans: List[Tuple[int, int, int]], bucket_elems: int) -> List[Tuple[int, int, int]]:
    """layer_spans: (layer_idx, offset, count) for the trainable layers' weight(+bias) slices,
    ascending and contiguous.  Returns buckets (trigger_layer_idx, offset, count), in the
    order they become ready during backward (highest layers first)."""
    buckets: List[Tuple[int, int, int]] = []
    cur_end: Optional[int] = None
    cur_start = 0
    trigger = -1
    for idx, off, cnt in reversed(layer_spans):
        if cur_end is None:
            cur_end = off + cnt
        cur_start = off
        trigger = idx
        if cur_end - cur_start >= bucket_elems:
            buckets.append((trigger, cur_start, cur_end - cur_start))
            cur_end = None
    if cur_end is not None:
        buckets.append((trigger, cur_start, cur_end - cur_start))
    return buckets


class GradientAllReduce:
    """wire: "f32" (default) or "bf16" -- the bucket is converted to bf16 for the exchange and the reduced
    result converted back into the f32 gradient arena (stage 2: 247 -> 123 MB per step on the links; each
    rank's contribution is rounded to 8 significant bits, the sum is what RCCL's bf16 reduction gives).
    algo: "allreduce" (default) or "rs_ag" -- reduce-scatter + all-gather of the (padded) bucket: on the
    fully connected xGMI mesh every link then carries 2/world of the bucket instead of a ring's 2(world-1)/world
    per link in sequence.  Unmeasured on hardware until a multi-GPU node runs bench.py."""

    def __init__(self, net, process_group=None, bucket_mb: float = 12.0, wire: str = "f32", algo: str = "allreduce",
                 inlist: bool = False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if wire not in ("f32", "bf16") or algo not in ("allreduce", "rs_ag"):
            raise ValueError("wire must be f32|bf16, algo allreduce|rs_ag")
        self.net = net
        self.pg = process_group
        self.wire, self.algo = wire, algo
        self.world_size = dist.get_world_size(process_group)
        # inlist: the exchange is a COMMAND of the recorded step (csrc/comm.hip): a communicator owned by the kernel
        # library, created once from an id rank 0 makes and torch.distributed hands round; the step's slices (the
        # optimizer's arena slices, net._plan_opt_chunks) are reduced on the list's exchange lane and swept right behind
        # their collective -- the list is never cut, nothing returns to Python between launches
        self.inlist = bool(inlist)
        self.comm = None
        self._stage_inlist = {}
        if self.inlist:
            from . import lib as L
            rank = dist.get_rank(process_group)
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0

            def exchange(ident):
                box = [ident]
                dist.broadcast_object_list(box, src=src, group=process_group)
                return box[0]
            torch.cuda.set_device(net.device)
            self.comm = L.Comm(rank, self.world_size, exchange)
        spans = []
        for l in net.layers:
            if l.lock:
                continue
            o, c = net.arena_slices["yolo/convolutional%d/weights" % l.idx]
            if l.kind == "lin":
                ob, cb = net.arena_slices["yolo/convolutional%d/biases" % l.idx]
                assert ob == o + c
                c += cb
            spans.append((l.idx, o, c))
        for (_, o, c), (_, o2, _) in zip(spans, spans[1:]):
            assert o + c == o2, "arena must be contiguous in layer order"
        self.buckets = plan_buckets(spans, int(bucket_mb * (1 << 20) / 4))
        # a bucket is final when every layer whose slice lies inside it has enqueued its weight
        # gradient -- the backward pass need not visit layers in strictly descending order
        self.bucket_of = {}
        self.members = []
        for bi, (_, bo, bc) in enumerate(self.buckets):
            mem = {idx for idx, o, c in spans if bo <= o and o + c <= bo + bc}
            self.members.append(mem)
            for idx in mem:
                self.bucket_of[idx] = bi
        assert sum(len(m) for m in self.members) == len(spans)
        self.by_trigger = {t: (o, c) for t, o, c in self.buckets}   # kept for introspection
        self.tail = (net.n_decay, net.n_params - net.n_decay)   # gamma/beta gradients
        self.works = []
        self._done = [set() for _ in self.buckets]
        # exchange buffers: one per bucket (+ the tail), padded to a multiple of the world size for the
        # reduce-scatter variant; allocated once (nothing is allocated inside the step)
        self._stage = {}
        if wire == "bf16" or algo == "rs_ag":
            dt = torch.bfloat16 if wire == "bf16" else torch.float32
            for key, (o, c) in list(enumerate((o, c) for _, o, c in self.buckets)) + [("tail", self.tail)]:
                if c > 0:
                    pad = -(-c // self.world_size) * self.world_size
                    self._stage[key] = torch.zeros(pad, dtype=dt, device=net.grad_arena.device)
        self._pending = []     # (key, offset, count) to copy back after the exchange
        # trace (set to a list): finish() brackets every wait with events on the compute stream -- the time the
        # optimizer's stream sits behind each bucket's collective ("exposed wait"; 0 when the exchange hid behind
        # the backward pass).  bench.py --gpus N reports it so that a first multi-GPU run explains itself.
        self.trace = None

    def begin_step(self) -> None:
        self.works = []
        self._pending = []
        self._done = [set() for _ in self.buckets]

    def completes_bucket(self, layer) -> Optional[int]:
        """record that `layer` is done; returns the bucket index this completes, if any"""
        bi = self.bucket_of.get(layer.idx)
        if bi is None:
            return None
        self._done[bi].add(layer.idx)
        return bi if self._done[bi] == self.members[bi] else None

    def _exchange(self, key, o: int, c: int) -> None:
        """sum grad_arena[o:o+c] over the ranks (async; finish() waits and copies back)"""
        if os.environ.get("DISYOLO_DP_DRY") == "1":   # timing probe: protocol without the collective
            return
        g = self.net.grad_arena[o:o + c]
        if not self._stage:
            self.works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            return
        st = self._stage[key]
        st[:c].copy_(g)                                   # f32 -> wire dtype, on the producing stream
        if self.algo == "rs_ag":
            n = st.numel() // self.world_size
            r = dist.get_rank(self.pg)
            shard = st[r * n:(r + 1) * n]
            w1 = dist.reduce_scatter_tensor(shard, st, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            w1.wait()                                     # stream-ordered: the gather reads the reduced shard
            self.works.append(dist.all_gather_into_tensor(st, shard, group=self.pg, async_op=True))
        else:
            self.works.append(dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self._pending.append((key, o, c))

    def exchange_inlist(self, key, o: int, c: int) -> None:
        """grad_arena[o:o+c] = its sum over the ranks, as commands of the current lane (or launches on the current
        stream outside a recording); wire / algo as the cut path's"""
        from . import lib as L
        if os.environ.get("DISYOLO_DP_DRY") == "1":
            return
        g = self.net.grad_arena[o:o + c]
        if self.wire == "f32" and self.algo == "allreduce":
            self.comm.allreduce(g)
            return
        st = self._stage_inlist.get(key)
        if st is None:
            pad = -(-c // self.world_size) * self.world_size
            st = torch.zeros(pad, dtype=torch.bfloat16 if self.wire == "bf16" else torch.float32, device=g.device)
            self._stage_inlist[key] = st
        if self.wire == "bf16":
            L.cast_f32_bf16(g, st)
        else:
            L.copy2d_f32(g, st, 1, c, c, c)
        if self.algo == "rs_ag":
            self.comm.reduce_scatter(st)
            self.comm.all_gather(st)
        else:
            self.comm.allreduce(st)
        if self.wire == "bf16":
            L.cast_bf16_f32(st, g)
        else:
            L.copy2d_f32(st, g, 1, c, c, c)

    def describe(self):
        """what bench.py prints as dp_exchange when the exchange is in the list (nothing to wait for on the host)"""
        net = self.net
        if net.opt_chunks is None:
            net._plan_opt_chunks()
        esz = 2 if self.wire == "bf16" else 4
        sizes = [ch["cnt"] * esz / 1e6 for ch in reversed(net.opt_chunks)] + [self.tail[1] * esz / 1e6]
        return {"mode": "commands of the recorded step (RCCL from the kernel library, lane %d)" % 3,
                "collectives_per_step": len(sizes), "bucket_mb": [round(x, 2) for x in sizes], "wire": self.wire,
                "algo": self.algo, "list_cuts": 0}

    def fire(self, bi: int) -> None:
        _, o, c = self.buckets[bi]
        self._exchange(bi, o, c)

    def on_layer_done(self, layer) -> None:
        bi = self.completes_bucket(layer)
        if bi is not None:
            self.fire(bi)

    def finish(self) -> None:
        o, c = self.tail
        if c > 0:
            self._exchange("tail", o, c)
        if self.trace is not None:
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(self.works) + 1)]
            evs[0].record()
            for i, w in enumerate(self.works):
                w.wait()
                evs[i + 1].record()
            self.trace.append(evs)
        else:
            for w in self.works:
                w.wait()
        for key, o, c in self._pending:                   # wire dtype -> the f32 arena
            self.net.grad_arena[o:o + c].copy_(self._stage[key][:c])
        self.works = []
        self._pending = []


    def trace_summary(self):
        """mean exposed wait per collective in microseconds, in the order finish() waits for them (the buckets in
        firing order, the gamma/beta tail last); call after a device synchronize"""
        if not self.trace:
            return None
        n = min(len(e) for e in self.trace) - 1
        out = []
        for i in range(n):
            out.append(round(sum(e[i].elapsed_time(e[i + 1]) for e in self.trace) / len(self.trace) * 1e3, 1))
        esz = 2 if self.wire == "bf16" else 4                  # bytes per element on the links
        sizes = [c * esz / 1e6 for _, _, c in self.buckets] + [self.tail[1] * esz / 1e6]
        return {"exposed_wait_us": out, "bucket_mb": [round(x, 2) for x in sizes[:n]], "steps": len(self.trace),
                "wire": self.wire, "algo": self.algo}


def broadcast_parameters(net, src: int = 0, process_group=None) -> None:
    """make every rank start from rank `src`'s variables (weights, BN statistics)"""
    for t in [net.arena] + [p for n, p in net.params.items() if p.data_ptr() < net.arena.data_ptr()
                            or p.data_ptr() >= net.arena.data_ptr() + net.arena.numel() * 4]:
        dist.broadcast(t, src=src, group=process_group)
    if not getattr(net, "plan_only", False):
        net.refresh_weights()


def enable_data_parallel(net, process_group=None, bucket_mb: float = 12.0, wire: Optional[str] = None,
                         algo: Optional[str] = None, broadcast: bool = True, sync_bn: bool = False,
                         inlist: Optional[bool] = None) -> GradientAllReduce:
    """wire / algo default to DISYOLO_DP_WIRE / DISYOLO_DP_ALGO (f32 / allreduce).  ``broadcast``: every
    rank starts from rank 0's variables (the reference has one process, hence one initialisation).
    ``sync_bn``: batch-norm statistics (forward moments and the two backward sums) over all ranks' batches,
    so that N ranks x b images train like one process with N*b images.
    ``inlist``: the exchange as commands of the recorded step (RCCL called from the kernel library, no list cuts);
    default: on for a CUDA net over an RCCL ("nccl") process group unless DISYOLO_DP_INLIST=0, off otherwise (gloo
    groups and plan-only CPU nets keep the cut list: torch.distributed issues their collectives)."""
    if getattr(net, "pair", False):
        # the pair step alternates two single-GPU lists and its eager form has no exchange point
        raise DisyoloError("backbone_pair is a single-GPU option: build the net without it for data parallelism")
    wire = wire or os.environ.get("DISYOLO_DP_WIRE", "f32")
    algo = algo or os.environ.get("DISYOLO_DP_ALGO", "allreduce")
    if inlist is None:
        inlist = (os.environ.get("DISYOLO_DP_INLIST", "1") != "0" and net.device.type == "cuda"
                  and not getattr(net, "plan_only", False) and dist.get_backend(process_group) == "nccl")
    # the in-launch batch norm (launches whose blocks wait for each other) is a single-GPU-process feature as shipped: beside RCCL's
    # own persistent kernels, which wait for OTHER ranks, its residency argument has never been exercised on more than one GPU
    # (no multi-GPU box in any round) -- more than one rank runs the separate batch-norm launches
    if dist.get_world_size(process_group) > 1 and getattr(net, "bn_inkernel", False) and not getattr(net, "plan_only", False):
        net.bn_inkernel = False
        if any(l.fused_fwd for l in net.layers):
            net._apply_tiles()
    net.dp = GradientAllReduce(net, process_group, bucket_mb, wire, algo, inlist=inlist)
    if broadcast and net.dp.world_size > 1:
        broadcast_parameters(net, 0, process_group)
    if sync_bn:
        # batch-norm statistics over the global batch (optional, default local): two all-reduces of
        # [C,2] f64 sums per trainable batch-norm layer and step, on the forward / backward critical chain
        net.enable_sync_bn()
    return net.dp
