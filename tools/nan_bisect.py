#!/usr/bin/env python3
"""Stage-2 (or stage-1) training at the BASELINE size, loss printed every step, and -- at the first
non-finite loss -- the first tensor of the step that is non-finite.  Run one process per switch
setting (the DISYOLO_* switches are read when the step is recorded):

    DISYOLO_BN_FUSE=0 python tools/nan_bisect.py --stage 2 --steps 60
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

import disyolo_amd  # noqa: F401
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch


def first_bad(net):
    """name of the first non-finite tensor in forward order, then backward order"""
    for l in net.layers:
        for nm in ("raw", "act"):
            t = getattr(l, nm)
            if t is not None and not bool(torch.isfinite(t.float()).all()):
                return "fwd layer %d %s" % (l.idx, nm)
        for nm in ("scale", "shift", "mean", "rstd"):
            t = getattr(l, nm)
            if t is not None and not bool(torch.isfinite(t).all()):
                return "fwd layer %d %s" % (l.idx, nm)
    for l in net.backward_order():
        for nm in ("grad", "dx", "dw", "dgamma", "dbeta", "dbias"):
            t = getattr(l, nm)
            if t is not None and not bool(torch.isfinite(t.float()).all()):
                return "bwd layer %d %s" % (l.idx, nm)
    if not bool(torch.isfinite(net.arena).all()):
        return "arena"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", type=int, default=2)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=576)
    ap.add_argument("--autotune", default="on")
    ap.add_argument("--tune-cache", default=None)
    ap.add_argument("--mode", default="program", choices=("program", "eager"))
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    net = YOLONet(training=True, device=dev, image_size=a.size, batch_size=a.batch, stage=a.stage, seed=0)
    net.set_batch(synthetic_batch(a.batch, a.size, seed=a.seed))
    torch.manual_seed(a.seed)
    if a.autotune == "on":
        picks = net.autotune(cache=a.tune_cache)
        print("picks", json.dumps({json.dumps(list(k)): v for k, v in picks.items() if v}), flush=True)
    net.shuffle_seed = a.seed
    if a.mode == "program":
        net.build_program()
    losses = []
    bad = None
    for i in range(a.steps):
        loss = float(net.train_step(None).cpu())
        s = net.summaries()
        losses.append(loss)
        print("%s step %3d total %.4f  obj %.3f noobj %.3f cls %.3f xy %.3f wh %.3f mask %.4f rois %s"
              % (a.tag, i, loss, s["object_loss"], s["noobject_loss"], s["class_loss"], s["xy_loss"], s["wh_loss"],
                 s["mask_loss"], net.roi_count.cpu().tolist()), flush=True)
        if not np.isfinite(loss):
            bad = first_bad(net)
            print("%s NON-FINITE at step %d; first bad tensor: %s" % (a.tag, i, bad), flush=True)
            print("rois:", net.rois.cpu().tolist()[:2], flush=True)
            break
    print(json.dumps({"tag": a.tag, "env": {k: v for k, v in os.environ.items() if k.startswith("DISYOLO_")},
                      "first": losses[0], "last": losses[-1], "n": len(losses), "bad": bad}))


if __name__ == "__main__":
    main()
