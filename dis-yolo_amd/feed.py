"""Host -> device input feed overlapped with the training step.

The reference hands every batch to ``sess.run`` as numpy arrays (``feed_dict``, train_yolo3_mask.py:146-149,216):
the copy to the device is part of the step.  At B = 8, 576x576 that is 90 MB (images f32 32 MB, instance masks 53 MB,
targets and boxes 5 MB) = 1.9 ms over PCIe against a 4.5 ms step when done synchronously.  ``HostFeeder`` keeps two
device staging sets: while step t runs, batch t+1 is copied from pinned host memory on a copy stream; at the start
of step t+1 the compute stream waits for that copy's event and moves the staging set into the network's input
buffers device-to-device (40 us).  Measured (tools/feed_rate.py, pinned host batches, round 2): stage 1 1828 img/s with resident inputs, 1294 with a
synchronous feed, 1778 with this one; stage 2 772 / 649 / 762.  Round 6: on a net whose step is the pipelined one the feeder hands
over the labels of batch t with the images of batch t + 1 and moves them on the net's feed stream (``YOLONet.feed_context``);
2074 img/s against 2180 resident and 1438 synchronous.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

KEYS = ("images", "clip_window", "true_boxes", "true_masks", "yolo1", "yolo2", "yolo3")


class HostFeeder:
    def __init__(self, net):
        self.net = net
        self.dev = net.device
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.staging = [None, None]
        self.pinned = [None, None]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.consumed = [torch.cuda.Event(), torch.cuda.Event()]
        for e in self.consumed:
            e.record()
        self._primed = False   # (pipelined nets: the first batch's backbone pass has been run)
        self.head = 0          # staging set the next submit() fills
        self.pending = []      # sets filled and not yet consumed, oldest first

    def _as_host(self, k: int, batch: Dict) -> Dict[str, torch.Tensor]:
        out = {}
        for key in KEYS:
            t = torch.as_tensor(batch[key])
            if key == "true_masks":
                t = t.to(torch.uint8)
            elif t.dtype != torch.float32:
                t = t.to(torch.float32)
            t = t.contiguous()
            if t.is_pinned():                          # the caller's loader already produces page-locked batches
                out[key] = t
                continue
            # pageable memory: one extra host copy into a page-locked buffer (10-20 ms for 90 MB on one core --
            # a loader that wants the overlap writes its batches into pinned tensors itself)
            if self.pinned[k] is None or key not in self.pinned[k] or self.pinned[k][key].shape != t.shape:
                self.pinned[k] = self.pinned[k] or {}
                self.pinned[k][key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
            # (the pinned buffer of set k was last read by the copy whose `ready` event the compute stream waited for)
            self.pinned[k][key].copy_(t)
            out[key] = self.pinned[k][key]
        return out

    def submit(self, batch: Dict) -> None:
        """queue one host batch (numpy arrays or CPU tensors, the seven feed_dict entries); at most two may be in flight"""
        if len(self.pending) >= 2:
            raise RuntimeError("HostFeeder: two batches are already in flight; call step() first")
        k = self.head
        self.consumed[k].synchronize()                 # host buffer / staging set k free again (never waits in steady state)
        host = self._as_host(k, batch)
        if self.staging[k] is None:
            self.staging[k] = {key: torch.empty(v.shape, dtype=v.dtype, device=self.dev) for key, v in host.items()}
        # (a pipelined net: its own feed stream carries the uploads too -- one probed stream, never parked on another's event)
        cs = self.copy_stream
        if getattr(self.net, "_pipe_in", None) is not None and getattr(self.net, "feed_stream", None) is not None:
            cs = self.net.feed_stream
        with torch.cuda.stream(cs):
            for key, v in host.items():
                self.staging[k][key].copy_(v, non_blocking=True)
            self.ready[k].record(cs)
        self.pending.append(k)
        self.head = 1 - k

    def step(self, det_thresh: Optional[float] = None, want_loss: bool = True):
        """train on the oldest submitted batch"""
        if not self.pending:
            raise RuntimeError("HostFeeder: no batch submitted")
        k = self.pending.pop(0)
        net = self.net
        if getattr(net, "_progs", None) is not None and getattr(net, "_pipe_in", None) is not None:
            # the pipelined step (YOLONet.build_program(pipeline_backbone=True)): labels of THIS batch, images of the NEXT
            # submitted one (whose backbone pass this step runs); the move into the net's input set happens on the net's
            # feed stream, beside the step that is still running.  The last step of a loop has no next batch: its own images
            # go in again (that backbone pass is not used by anybody).
            k1 = self.pending[0] if self.pending else k
            if not self._primed:
                torch.cuda.current_stream().wait_event(self.ready[k])
                net.prime_pipeline(self.staging[k]["images"], self.staging[k]["clip_window"])
                self._primed = True
            mixed = dict(self.staging[k])
            mixed["images"] = self.staging[k1]["images"]
            with net.feed_context():
                fs = torch.cuda.current_stream()
                fs.wait_event(self.ready[k])
                fs.wait_event(self.ready[k1])
                net.set_batch(mixed)
                self.consumed[k].record(fs)
        else:
            torch.cuda.current_stream().wait_event(self.ready[k])
            net.set_batch(self.staging[k])             # device-to-device on the compute stream
            self.consumed[k].record()
        if det_thresh is None:
            return self.net.train_step(None, want_loss=want_loss)
        return self.net.train_step(None, det_thresh=det_thresh, want_loss=want_loss)

