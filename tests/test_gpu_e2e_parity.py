"""North-star parity sentence, end to end: "boxes, class logits and assembled masks match the reference on identical
inputs within a stated fp tolerance" -- the HIP inference path (bf16 storage) against the f32 oracle on a TRAINED net
at the BASELINE sizes, detection by detection (calculate_test_map.py:218-266 consumes exactly these arrays).

The stated tolerance (measured in profiles/r04_e2e_parity.json):
  * >= 90 % of the f32 oracle's detections are found with the same class and box IoU >= 0.75, every one of them
    that is not within 0.1 of the score threshold bar at most one per batch (an NMS survivor can flip between two
    near-duplicate candidates);
  * >= 65 % with box IoU >= 0.9 (a 1 % logit error is 5 % of a box side through anchor * exp(t));
  * pairs matched at IoU >= 0.9: |score difference| <= 0.12 for all but at most ONE pair of the batch (<= 0.3: a detection whose
    confidence logit sits where the sigmoid is steep -- the net was fitted THROUGH the bf16 forward pass, the f32 oracle evaluating
    the same variables sees confidence logits ~1 lower at the detected cells; how many such detections a net has depends on how far
    its 3,000-step overfit got, the bf16-emulating oracle shows 0.18 on the same variables), median <= 0.06; mask IoU after "> 0.5"
    >= 0.8 each, >= 0.93 on average;
  * and the yardstick: the oracle itself with every stored tensor rounded to bf16 reproduces the f32 oracle no
    better than the HIP path does (within 0.1 of its IoU-0.75 rate) -- the gap is bf16 storage, not the kernels.
The oracle is pinned by hand KATs only (TF 1.x cannot run here): "parity unpinned" applies to this file too."""
import json
import os

import pytest
import torch

import e2e_parity as E

pytestmark = pytest.mark.gpu
STEPS = 3000
THR = 0.25            # cfg.OBJ_THRESHOLD, the value evaluate() passes


@pytest.fixture(scope="module")
def trained(dev):
    out = {}
    for S, B in ((576, 8), (832, 4)):
        batch = E.painted_batch(B, S, seed=5)
        sd, curve = E.train_overfit(dev, batch, B, S, STEPS)
        assert curve[-1] < 0.01 * curve[0] and all(c == c for c in curve), curve      # the recorded step really trained it
        out[S] = (batch, sd)
    return out


def _gate(r, yardstick=None):
    v = r["vs_f32"]
    assert v["ref_detections"] >= 3, v                                   # a trained detector, not an empty comparison
    assert v["reproduced_iou75_frac"] >= 0.9, v
    assert v["confident_reproduced_iou75"] >= v["confident_ref"] - 1, v
    assert v["reproduced_iou90_frac"] >= 0.65, v
    assert v["score_absdiff_2nd"] <= 0.12 and v["score_absdiff_max"] <= 0.3 and v["score_absdiff_median"] <= 0.06, v
    assert v["mask_iou_min"] >= 0.8 and v["mask_iou_mean"] >= 0.93, v
    assert v["hip_unmatched"] <= max(1, 0.35 * v["hip_detections"]), v
    assert max(v["logit_rel_l2"][:3]) < 3e-2 and v["logit_rel_l2"][3] < 0.2, v
    if yardstick is not None:
        assert v["reproduced_iou75_frac"] >= yardstick["reproduced_iou75_frac"] - 0.1, (v, yardstick)


@pytest.mark.parametrize("S,Bi", [(576, 8), (576, 1), (832, 4)])
def test_detections_and_masks_agree_with_the_f32_oracle(dev, trained, S, Bi):
    batch, sd = trained[S]
    r = E.compare(dev, sd, batch["images"][:Bi], batch["clip_window"][:Bi], S, THR, with_bf16_oracle=(Bi > 1))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(r, open("gpurun_out/e2e_parity_S%d_B%d.json" % (S, Bi), "w"), indent=1)
    _gate(r, r.get("bf16_vs_f32"))
    if "vs_bf16" in r:
        # against the bf16-emulating oracle (same roundings, another summation order): the same bars
        q = r["vs_bf16"]
        assert q["reproduced_iou75_frac"] >= 0.85 and q["mask_iou_mean"] >= 0.93, q
