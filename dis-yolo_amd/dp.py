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
    def __init__(self, net, process_group=None, bucket_mb: float = 12.0):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.net = net
        self.pg = process_group
        self.world_size = dist.get_world_size(process_group)
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

    def begin_step(self) -> None:
        self.works = []
        self._done = [set() for _ in self.buckets]

    def completes_bucket(self, layer) -> Optional[int]:
        """record that `layer` is done; returns the bucket index this completes, if any"""
        bi = self.bucket_of.get(layer.idx)
        if bi is None:
            return None
        self._done[bi].add(layer.idx)
        return bi if self._done[bi] == self.members[bi] else None

    def fire(self, bi: int) -> None:
        _, o, c = self.buckets[bi]
        if os.environ.get("DISYOLO_DP_DRY") == "1":   # timing probe: protocol without the collective
            return
        self.works.append(dist.all_reduce(self.net.grad_arena[o:o + c], op=dist.ReduceOp.SUM, group=self.pg,
                                          async_op=True))

    def on_layer_done(self, layer) -> None:
        bi = self.completes_bucket(layer)
        if bi is not None:
            self.fire(bi)

    def finish(self) -> None:
        o, c = self.tail
        if c > 0 and os.environ.get("DISYOLO_DP_DRY") != "1":
            self.works.append(dist.all_reduce(self.net.grad_arena[o:o + c], op=dist.ReduceOp.SUM, group=self.pg,
                                              async_op=True))
        for w in self.works:
            w.wait()
        self.works = []


def broadcast_parameters(net, src: int = 0, process_group=None) -> None:
    """make every rank start from rank `src`'s variables (weights, BN statistics)"""
    for t in [net.arena] + [p for n, p in net.params.items() if p.data_ptr() < net.arena.data_ptr()
                            or p.data_ptr() >= net.arena.data_ptr() + net.arena.numel() * 4]:
        dist.broadcast(t, src=src, group=process_group)
    net.refresh_weights()


def enable_data_parallel(net, process_group=None, bucket_mb: float = 12.0) -> GradientAllReduce:
    net.dp = GradientAllReduce(net, process_group, bucket_mb)
    return net.dp
