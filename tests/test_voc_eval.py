"""The mAP evaluator against golden vectors produced by the reference's own
utils/voc_eval_mask.py (tools/make_golden.py) -- the one pinned piece of this project."""
import json
import os

import numpy as np

from disyolo_amd.voc_eval import compute_overlaps_masks, voc_ap, voc_eval

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "voc_eval.json")))


def test_voc_ap_matches_reference():
    for c in G["ap_cases"]:
        assert abs(voc_ap(c["rec"], c["prec"], False) - c["ap"]) < 1e-12
        assert abs(voc_ap(c["rec"], c["prec"], True) - c["ap07"]) < 1e-12


def test_mask_overlaps_match_reference():
    o = G["overlaps"]
    got = compute_overlaps_masks(np.asarray(o["m1"], np.float32), np.asarray(o["m2"], np.float32))
    np.testing.assert_allclose(got, np.asarray(o["iou"]), rtol=1e-6)
    assert compute_overlaps_masks(np.zeros((4, 4, 0)), np.zeros((4, 4, 2))).shape == (0, 2)


def test_voc_eval_matches_reference_cases():
    for case in G["cases"]:
        recs = {n: [{"classid": o["classid"], "difficult": o["difficult"], "mask": np.asarray(o["mask"], np.uint8)}
                    for o in objs] for n, objs in case["recs"].items()}
        dets = [{"imageid": d["imageid"], "score": d["score"], "classid": d["classid"],
                 "mask": np.asarray(d["mask"], np.uint8)} for d in case["dets"]]
        for key, want in case["results"].items():
            cid, use07 = (int(v) for v in key.split("_"))
            got = voc_eval([d for d in dets if d["classid"] == cid], recs, case["names"], cid, 0.5, bool(use07))
            np.testing.assert_allclose(np.asarray(got, float), np.asarray(want, float), rtol=1e-12, equal_nan=True)


def test_known_answer_three_detections_two_gt():
    """SURVEY F3 hand calculation: 3 detections / 2 GT -> recall 1.0, precision 2/3, AP 5/6."""
    def box(y, x):
        m = np.zeros((10, 10), np.uint8)
        m[y:y + 4, x:x + 4] = 1
        return m
    recs = {"a": [{"classid": 0, "difficult": 0, "mask": box(0, 0)}, {"classid": 0, "difficult": 0, "mask": box(5, 5)}]}
    dets = [{"imageid": "a", "score": 0.9, "mask": box(0, 0)}, {"imageid": "a", "score": 0.8, "mask": box(0, 5)},
            {"imageid": "a", "score": 0.7, "mask": box(5, 5)}]
    r, p, ap = voc_eval(dets, recs, ["a"], 0)
    assert abs(r - 1.0) < 1e-12 and abs(p - 2.0 / 3.0) < 1e-12 and abs(ap - 5.0 / 6.0) < 1e-12
    assert voc_eval([], recs, ["a"], 0) == (0.0, 0.0, 0.0)


def test_voc_eval_with_precomputed_overlap_rows_equals_the_mask_path():
    """round 6: evaluate.MAP.collect hands voc_eval the IoU row of every detection (taken on the GPU from exact pixel counts)
    instead of its mask.  Same golden cases: rows computed the way collect computes them (counts, f32 division) give the reference's
    recall / precision / AP, and equal the mask path's bit for bit."""
    for case in G["cases"]:
        recs = {n: [{"classid": o["classid"], "difficult": o["difficult"], "mask": np.asarray(o["mask"], np.uint8)}
                    for o in objs] for n, objs in case["recs"].items()}
        for c in sorted({d["classid"] for d in case["dets"]}):
            dets = [{"imageid": d["imageid"], "score": d["score"], "mask": np.asarray(d["mask"], np.uint8)}
                    for d in case["dets"] if d["classid"] == c]
            rows = []
            for d in dets:
                objs = [o for o in recs[d["imageid"]] if o["classid"] == c]
                if not objs:
                    rows.append({"imageid": d["imageid"], "score": d["score"], "ov": np.zeros(0, np.float32)})
                    continue
                g = np.stack([(o["mask"].astype(float) > 0.5).reshape(-1) for o in objs]).astype(np.float32)     # [ng, HW]
                m = (d["mask"].astype(float) > 0.5).reshape(-1).astype(np.float32)
                inter = g @ m
                union = m.sum() + g.sum(1) - inter
                rows.append({"imageid": d["imageid"], "score": d["score"], "ov": (inter / union).astype(np.float32)})
            names = case["names"]
            want = voc_eval(dets, recs, names, c, ovthresh=0.5)
            got = voc_eval(rows, recs, names, c, ovthresh=0.5)
            np.testing.assert_array_equal(np.asarray(got, float), np.asarray(want, float))     # (NaN where there is no ground truth: in both)
