#!/usr/bin/env python3
"""Tile tables bench.py loads by default (GPU box): the in-sequence autotuner with more passes than bench.py's
setup run takes (median of 9 instead of 3), one table per workload, written to gpurun_out/ (copy them to profiles/).

    python tools/make_tune_tables.py [train1] [train2] [infer] [train832] [train832fp8] [train1pair]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

import disyolo_amd  # noqa: F401
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
dev = torch.device("cuda:0")
what = sys.argv[1:] or ["train1", "train2", "infer", "train832", "train832fp8"]
for w in what:
    L.TUNED.clear()
    if w == "infer":
        B, S, name = 32, 576, "tune_infer_B32_576.json"
        net = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
        b = synthetic_batch(B, S, seed=1234)
        net._set_inputs(b["images"], b["clip_window"])
    else:
        stage = 2 if w == "train2" else 1
        B, S = (4, 832) if "832" in w else (8, 576)
        dtype = "fp8" if w.endswith("fp8") else "bf16"
        pair = w.endswith("pair")
        name = "tune_train_B%d_%d_stage%d%s%s.json" % (B, S, stage, "" if dtype == "bf16" else "_fp8", "_pair" if pair else "")
        net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=stage, seed=0, dtype=dtype, backbone_pair=pair)
        net.set_batch(synthetic_batch(B, S, seed=1234))
        if pair:
            net.set_batch(synthetic_batch(B, S, seed=4321), 1)
        if dtype == "fp8":
            net.calibrate_fp8()
    path = os.path.join(OUT, name)
    if os.path.exists(path):
        os.remove(path)
    # training workloads: the persistent streaming kernel (tile 20: one 162-KB-LDS block per CU) is timed on ONE lane by the
    # tuner and wins there, but in the two-lane step it cannot share a CU with the side lane's resident blocks and loses
    # (tools/instep_tune.py, round 4: 4.26 -> 4.21 ms with the 128x64 tile instead) -- not a candidate for training tables
    cands = tuple(c for c in L.TUNE_CANDIDATES if not (w != "infer" and (c & 0xff) == 20))
    picks = net.autotune(reps=9, cache=path, candidates=cands)
    print(name, "%d shapes, %d off the heuristic" % (len(picks), sum(1 for v in picks.values() if v)), flush=True)
    del net
    torch.cuda.empty_cache()
