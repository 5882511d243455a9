"""CPU-side logic of the product: topology table, synthetic batches, bucket planning."""
import numpy as np
import torch

import disyolo_oracle as O
from disyolo_amd import config as cfg
from disyolo_amd.dp import plan_buckets
from disyolo_amd.net import build_topology, var_name
from disyolo_amd.synth import synthetic_batch
from disyolo_amd import lib as L


def test_topology_table_matches_oracle_restatement():
    layers = build_topology(3, 3)
    spec = O.layer_specs()
    assert len(layers) == 82
    for l in layers:
        cin, cout, k, s, kind = spec[l.idx]
        assert (l.cin, l.cout, l.k, l.stride, l.kind) == (cin, cout, k, s, kind), l.idx
    by = {l.idx: l for l in layers}
    # concat partners [skip, upsampled] (yolo/yolo3_net_pos.py:291,326,387,402)
    assert (by[61].src, by[61].src_up) == (43, 60) and (by[69].src, by[69].src_up) == (26, 68)
    assert (by[77].src, by[77].src_up) == (9, 76) and (by[80].src, by[80].src_up) == (4, 79)
    assert by[60].src == 57 and by[68].src == 65 and by[76].src == 73      # branch points
    assert all(by[i].shortcut == i - 2 for i in by if by[i].kind == "res")
    n = sum(l.k * l.k * l.cin * l.cout for l in layers)
    assert n == 61602208
    assert var_name(59, "biases") == "yolo/convolutional59/biases"


def test_same_pads_helper():
    assert L.same_pads(576, 3, 2) == (288, 0) and L.same_pads(576, 3, 1) == (576, 1) and L.same_pads(17, 3, 2) == (9, 1)


def test_synthetic_batch_matches_oracle_generator_and_assignment():
    a = synthetic_batch(2, 96, seed=5)
    b = O.synthetic_batch(2, 96, seed=5)
    for k in ("images", "true_boxes", "yolo1", "yolo2", "yolo3"):
        np.testing.assert_array_equal(a[k], b[k].numpy())
    np.testing.assert_array_equal(a["true_masks"], b["true_masks"])
    # every object cell carries a one-hot class and a normalised box
    for key, g in (("yolo3", 12), ("yolo2", 6), ("yolo1", 3)):
        t = a[key]
        assert t.shape == (2, g, g, 3, 8)
        obj = t[..., 4] == 1
        assert (t[..., 5:][obj].sum(-1) == 1).all()
        assert (t[..., :4][obj] > 0).all() and (t[..., :4][obj] <= 1).all()
    assert sum(int((a[k][..., 4] == 1).sum()) for k in ("yolo1", "yolo2", "yolo3")) >= 2


def test_bucket_plan_covers_arena_in_backward_order():
    spans = [(53, 0, 100), (54, 100, 900), (55, 1000, 100), (56, 1100, 900), (59, 2000, 30)]
    b = plan_buckets(spans, 500)
    # highest layers first, contiguous, disjoint, complete
    assert b[0][0] == 56 and b[0][1] == 1100 and b[0][1] + b[0][2] == 2030
    covered = sorted((o, o + c) for _, o, c in b)
    assert covered[0][0] == 0 and covered[-1][1] == 2030
    for (a0, a1), (b0, b1) in zip(covered, covered[1:]):
        assert a1 == b0
    triggers = [t for t, _, _ in b]
    assert triggers == sorted(triggers, reverse=True)
    assert plan_buckets(spans, 10 ** 9) == [(53, 0, 2030)]


def test_committed_bench_line_follows_the_contract():
    """the JSON line bench.py printed for the committed profile run carries every field the driver
    and the judge read (metric/unit/value..., roofline, cpu_baseline, config.workload)"""
    import json
    import os
    import glob
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    # the newest committed line of the current rounds (earlier rounds: profiles/archive/)
    cands = sorted(glob.glob(os.path.join(prof, "r[0-9][0-9]*_bench_stage1.json")))
    assert cands, "no committed bench line under profiles/"
    d = json.loads(open(cands[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and d["dtype"] == "bf16" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference")
    # value is consistent with the step time: images per step / seconds per step
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
