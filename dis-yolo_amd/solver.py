"""Training driver: the counterpart of ``train_yolo3_mask.py`` (class ``Solver`` :20-235, ``main`` :237-248).

Same loop as the reference: every step ``data.get()`` -> ``sess.run([total_loss, optimizer])``; every
SUMMARY_ITER steps the seven summary scalars; every 10 * SUMMARY_ITER steps a full validation sweep
(``net.evaluation`` per batch -> ``MAP.do_python_eval``) and a log line; every SAVE_ITER steps a TF-format
checkpoint ``model.ckpt-<step>`` of the ``yolo/convolutional{1..82}`` variables plus ``<step>map.npy``.

Differences, all deliberate and switchable:
  * the step is the recorded HIP program of ``YOLONet`` (one C call per step) instead of a TF session;
  * the learning rate.  In the reference the schedule of :130-141 is dead code: the optimizer captured
    1e-4 when the graph was built (:38,55) and the later assignments only change what is printed
    (SURVEY F6).  ``lr_schedule="faithful"`` (default) therefore trains at 1e-4 and prints the
    schedule's value like the reference; ``"intended"`` applies it (the optimizer kernel reads the rate
    from device memory, so the recorded step follows it);
  * summaries go to ``events.jsonl`` in the checkpoint directory (no TensorBoard writer here);
  * the loss of every step is accumulated like the reference does (:216-218), but FETCHED in blocks: the optimizer's
    finish files each step's total loss in a device ring (``YOLONet.step_losses``) and the loop reads it at the summary
    / checkpoint steps (at the latest every ``YOLONet.LOSS_RING`` steps).  Same values, same order of accumulation; the
    device is joined once per block instead of once per step, so the recorded step's overlapped tail
    (``build_program(overlap_tail=True)``: what bench.py times) is what a real run gets too.
"""
from __future__ import annotations

import datetime
import json
import os
import time
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import config as cfg
from .checkpoint import restore_net, save_net


class Timer:
    """Wall-clock bookkeeping for the log line of Solver.train: mean seconds per tic/toc interval and an ETA from the
    time since construction (the role utils/timer.py plays for train_yolo3_mask.py:143-195)."""

    def __init__(self):
        self._born = time.perf_counter()
        self._t0 = self._born
        self._sum, self._n = 0.0, 0

    def tic(self) -> None:
        self._t0 = time.perf_counter()

    def toc(self, count: int = 1) -> float:
        """``count`` intervals ended since tic() (a block of steps timed as one)"""
        self._sum += time.perf_counter() - self._t0
        self._n += count
        return self.average_time

    @property
    def average_time(self) -> float:
        return self._sum / max(self._n, 1)

    def remain(self, done: int, total: int) -> str:
        """h:mm:ss left if the remaining iterations go at the pace of the ``done`` ones so far"""
        left = 0 if done <= 0 else (time.perf_counter() - self._born) / done * max(total - done, 0)
        return str(datetime.timedelta(seconds=int(left)))


def scheduled_learning_rate(step: int) -> float:
    """the values train_yolo3_mask.py:130-141 assigns (and, in the reference, only prints)"""
    if step <= 10000:
        return 1e-3
    if step <= 20000:
        return 1e-4
    if step <= 25000:
        return 1e-5
    return 1e-6


class Solver(object):
    def __init__(self, net, data, evalu=None, val_data=None, output_dir: Optional[str] = None,
                 lr_schedule: str = "faithful", restore_weight: Optional[str] = None, stage: int = 1,
                 max_iter: Optional[int] = None, summary_iter: Optional[int] = None, save_iter: Optional[int] = None,
                 log: Callable[[str], None] = print, use_program: bool = True, shuffle_seed: Optional[int] = 20190530,
                 pipeline_backbone: Optional[bool] = None):
        """net: YOLONet(training=True); data: object with ``get()`` -> (images, true_masks, true_boxes, yolo_3,
        yolo_2, yolo_1, window) and attributes epoch / image_size / batch_size (utils/train_data.py:44-276);
        evalu: ``MAP``; val_data: object with ``get()`` -> (images [N,S,S,3], image ids, windows [N,4])
        (utils/val_data.py:23-34); restore_weight: checkpoint prefix (cfg.WEIGHTS_FILE in the reference)."""
        if lr_schedule not in ("faithful", "intended"):
            raise ValueError("lr_schedule must be 'faithful' or 'intended'")
        self.net, self.data, self.eval, self.val_data = net, data, evalu, val_data
        self.start_iter = 1
        self.max_iter = cfg.MAX_ITER if max_iter is None else max_iter
        self.summary_iter = cfg.SUMMARY_ITER if summary_iter is None else summary_iter
        self.save_iter = cfg.SAVE_ITER if save_iter is None else save_iter
        out = cfg.OUTPUT_DIR if output_dir is None else output_dir
        self.ckpt_dir = os.path.join(out, "checkpoint")
        self.loss_dir = os.path.join(out, "lossnp")
        self.ckpt_file = os.path.join(self.ckpt_dir, "model.ckpt")
        os.makedirs(self.ckpt_dir, exist_ok=True)
        os.makedirs(self.loss_dir, exist_ok=True)
        self.save_cfg()
        self.lr_schedule = lr_schedule
        self.learning_rate = 1e-4                 # :38 -- what the optimizer uses in "faithful" mode
        self.log = log
        self.use_program = use_program
        # stage 1: the locked backbone of the NEXT batch runs on a third lane while the heads / losses / backward / Adam of the
        # current one run (YOLONet.build_program(pipeline_backbone=True): bit-identical variables, +5.9 % on one MI355X) -- the
        # loop then reads its data one batch ahead.  None = on where it applies (a locked prefix, the list executor, no pair).
        self.pipeline_backbone = pipeline_backbone
        self.global_step = 0
        # tf.random_shuffle of the mask-loss RoIs every step (yolo/yolo3_net_pos.py:781-782): on the device, seeded
        # (a net that already has a seed -- or injected permutations with shuffle_seed=None -- keeps it)
        if net.shuffle_seed is None and shuffle_seed is not None:
            if getattr(net, "_prog", None) is not None:
                # the recorded step was built without the device-side shuffle: setting the seed now would change nothing
                # and the RoI order would stay fixed for the whole run
                raise ValueError("Solver: the net's step was recorded (build_program) before a shuffle seed was set; set "
                                 "net.shuffle_seed before build_program(), or pass shuffle_seed=None to keep a fixed RoI order")
            net.shuffle_seed = shuffle_seed
        self.events = open(os.path.join(self.ckpt_dir, "events.jsonl"), "a")
        self.log("*** Train variables ***")
        for i, name in enumerate(net.trainable_names()):
            self.log("  param {:3}: {:15}   {}".format(i, str(tuple(net.params[name].shape)), name))
        if restore_weight:
            self.log("Restoring weights from: " + restore_weight)
            # stage 1: the include list of :75-107 (ignore_missing_vars); stage 2: everything (:111)
            restore_net(net, restore_weight, stage1_include=(stage == 1))
        net.learning_rate = self.learning_rate

    def save_cfg(self):
        """train_yolo3_mask.py:229-235"""
        with open(os.path.join(self.ckpt_dir, "config.txt"), "w") as f:
            for key in sorted(cfg.__dict__.keys()):
                if key[0].isupper():
                    f.write("{}: {}\n".format(key, cfg.__dict__[key]))

    def _feed(self):
        on_device = getattr(self.data, "get_device", None)
        if on_device is not None:
            return on_device()       # (train_data.defect_train: every array already on the GPU, nothing waits for the device)
        images, true_masks, true_boxes, yolo_3, yolo_2, yolo_1, window = self.data.get()
        return {"images": images, "true_masks": true_masks, "true_boxes": true_boxes, "yolo3": yolo_3, "yolo2": yolo_2,
                "yolo1": yolo_1, "clip_window": window}

    def validate(self):
        """the sweep of :164-178; returns thresh_out[0]"""
        imagesval, imageidsval, window_vals = self.val_data.get()
        num_val, B = len(imageidsval), self.net.B
        if num_val % B:
            self.log("Please manually change the number of validation data.")      # :123-124
        detect = []
        for v in range(num_val // B):
            a, b = B * v, B * v + B
            det_boxes, det_masks = self.net.evaluation(imagesval[a:b], window_vals[a:b], [np.float32(cfg.OBJ_THRESHOLD)],
                                                       masks_on_device=True)
            detect.extend({"boxes": det_boxes[i], "masks": det_masks[i], "imname": imageidsval[a + i]} for i in range(B))
        return self.eval.do_python_eval(detect)[0]

    def train(self):
        load_timer, train_timer = Timer(), Timer()
        val_map = np.zeros((800, 9))
        epoch_loss = 0.0
        net = self.net
        pipe = False
        if self.use_program and net._prog is None:
            net.set_batch(self._feed_peek())
            if getattr(net, "pair", False):
                net.set_batch(self._pending, 1)      # (placeholder inputs while the two lists are recorded)
            pipe = self.pipeline_backbone
            if pipe is None:
                pipe = (net.use_side_lane and not getattr(net, "pair", False) and net._backbone_prefix() >= 2
                        and net.dtype == "bf16" and not getattr(net, "sync_bn", False))
            overlap = (not pipe and net.use_side_lane and not getattr(net, "pair", False) and (net.dp is None or net.dp.inlist))
            net.build_program(det_thresh=cfg.OBJ_THRESHOLD, overlap_tail=overlap, pipeline_backbone=bool(pipe))
        elif self.use_program:
            pipe = getattr(net, "_progs", None) is not None and not getattr(net, "pair", False)      # (the caller recorded the step)
        ahead = None

        def prime():
            """(re)compute the backbone pass of the batch the next step trains on: at the start, and after anything that ran
            another forward pass through the net's buffers (the validation sweep)"""
            net.prime_pipeline(ahead["images"], ahead["clip_window"])
        if pipe and self.start_iter <= self.max_iter:
            ahead = self._next_feed()
            prime()
        history = []
        first_pending = net.step_count       # the ring index of the first step whose loss has not been fetched yet
        first_step_count = first_pending
        n_pending = 0
        train_timer.tic()

        nonfinite_seen = [False]
        load_in_block = [0.0]      # data-loading seconds inside the block train_timer is timing (load_timer counts them too)

        def fetch():
            """the losses of the steps run since the last fetch: accumulated one by one, in order, like :218"""
            nonlocal epoch_loss, first_pending, n_pending
            if n_pending:
                if net.n_params == 0:
                    vals = [float(net.total_loss().cpu())] * n_pending      # (nothing trainable: no optimizer finish files a loss)
                else:
                    vals = net.step_losses(first_pending, n_pending)        # (joins the device)
                net.check_cluster_sync()
                for i, v in enumerate(vals):
                    if not np.isfinite(v) and not nonfinite_seen[0]:
                        # reported at the first fetch that sees it, with the step it happened at -- reported, not raised: a step
                        # whose mask loss meets a zero-area positive RoI IS NaN in the reference too, and the
                        # reference keeps training through it
                        nonfinite_seen[0] = True
                        self.log("non-finite total loss at step %d (noticed at step %d, when the losses were fetched)"
                                 % (self.start_iter + first_pending + i - first_step_count, self.global_step))
                    epoch_loss += float(v)
                    history.append(float(v))
                # the block's wall time minus what it spent loading data: speed and load stay separate figures, like
                # train_yolo3_mask.py:143-195's two timers
                train_timer._sum -= load_in_block[0]
                load_in_block[0] = 0.0
                train_timer.toc(n_pending)
                first_pending += n_pending
                n_pending = 0
            train_timer.tic()

        for step in range(self.start_iter, self.max_iter + 1):
            shown_lr = scheduled_learning_rate(step)
            if self.lr_schedule == "intended" and shown_lr != self.learning_rate:
                self.learning_rate = shown_lr
                net.learning_rate = shown_lr
            load_timer.tic()
            if getattr(net, "pair", False):
                # backbone_pair: the even step takes its own batch and the next one (the locked backbone runs on both),
                # the odd step takes nothing
                feed = (self._next_feed(), self._next_feed()) if net._parity_now() == 0 else None
            elif pipe:
                # the labels of this step's batch, the images of the next one (whose backbone pass this step runs beside its own
                # heads / backward); one batch is read ahead -- the batches a step trains on are the reference's, in its order
                # -- produced and copied on the net's feed stream, beside the step that is running (two input sets)
                with net.feed_context():
                    nxt = self._feed()
                    feed = dict(ahead)
                    feed["images"] = nxt["images"]
                    ahead = nxt
                    net.set_batch(feed)
                feed = None
            else:
                feed = self._next_feed()
            load_timer.toc()
            load_in_block[0] += time.perf_counter() - load_timer._t0
            net.train_step(feed, det_thresh=cfg.OBJ_THRESHOLD, want_loss=False)
            n_pending += 1
            self.global_step += 1
            if (step % self.summary_iter == 0 or step % self.save_iter == 0 or step == self.max_iter
                    or n_pending == net.LOSS_RING):
                fetch()
            if step % self.summary_iter == 0:
                summ = net.summaries()
                summ["step"] = step
                self.events.write(json.dumps(summ) + "\n")
                self.events.flush()
                if step % (self.summary_iter * 10) == 0 and self.eval is not None and self.val_data is not None:
                    thresh_out = self.validate()
                    if pipe and step < self.max_iter:
                        prime()          # (the sweep ran its images through the backbone's buffers)
                    record_loss = epoch_loss / self.save_iter
                    row = int(step / (self.summary_iter * 10)) - 1
                    if row < val_map.shape[0]:
                        val_map[row, :] = [step, getattr(self.data, "epoch", 0), record_loss] + list(thresh_out["AP"][:3]) + \
                            list(thresh_out["mAP"][:3])
                    self.log(("{} Epoch: {}, Step: {}, Image: {}, Batch: {}, Learning rate: {},"
                              " Loss: {:5.3f}, crack: {:5.3f}, spall: {:5.3f}, rebar: {:5.3f}, mAP50: {:5.3f},"
                              "\nSpeed: {:.3f}s/iter, Load: {:.3f}s/iter, Remain: {}").format(
                        datetime.datetime.now().strftime("%m/%d %H:%M:%S"), getattr(self.data, "epoch", 0), int(step),
                        getattr(self.data, "image_size", net.S), getattr(self.data, "batch_size", net.B),
                        round(shown_lr, 6), record_loss, thresh_out["AP"][0], thresh_out["AP"][1], thresh_out["AP"][2],
                        thresh_out["mAP"][2], train_timer.average_time, load_timer.average_time,
                        train_timer.remain(step, self.max_iter)))
                    epoch_loss = 0.0
                    train_timer.tic()
            if step % self.save_iter == 0:
                self.log("{} Saving checkpoint file to: {}".format(datetime.datetime.now().strftime("%m/%d %H:%M:%S"),
                                                                   self.ckpt_dir))
                save_net(net, "%s-%d" % (self.ckpt_file, step))                       # saver.save(..., global_step=step)
                np.save(os.path.join(self.loss_dir, str(step) + "map.npy"), val_map)
        return history

    # the first batch is needed once before the step can be recorded; it is then used as step 1's batch
    def _feed_peek(self):
        self._pending = self._feed()
        return self._pending

    def _next_feed(self):
        p = getattr(self, "_pending", None)
        if p is not None:
            self._pending = None
            return p
        return self._feed()
