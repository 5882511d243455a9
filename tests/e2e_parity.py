"""End-to-end agreement of the HIP inference path with the f32 oracle: boxes, class ids, scores and assembled
masks (calculate_test_map.py:218-266 -- what evaluate() consumes -- over yolo/yolo3_net_pos.py:517-628,862-938).

Test infrastructure (imported by tests/test_gpu_e2e_parity.py and tools/e2e_parity_report.py only).  The per-kernel
parity tests compare the detection filter and the mask assembly ON THE KERNELS' OWN LOGITS; this module answers the
question the north_star asks: how much of what the f32 graph detects does the bf16 path reproduce, end to end?

A randomly initialised net puts every score next to the threshold, where a bf16 rounding decides; so the net is first
trained (stage 1, the recorded HIP step) on ONE batch whose images carry the objects (class-coloured ellipses on a
noise background) until its detections are confident, then both paths run inference with the SAME f32 variables:
  HIP:    YOLONet(training=False).evaluation()     -- bf16 storage, f32 accumulation, fused launches, tuned or default tiles
  oracle: build_network (f32) -> interpret_output -> filter_detections -> val_test
"""
import numpy as np
import torch

import disyolo_oracle as O
from disyolo_amd import config as cfg
from disyolo_amd.net import YOLONet

CLASS_COLOUR = np.array([[0.9, 0.15, 0.15], [0.15, 0.9, 0.15], [0.15, 0.15, 0.9]], np.float32)


def painted_batch(B: int, S: int, seed: int):
    """O.synthetic_batch with its instances drawn INTO the images: background = noise around 0.5, every instance an
    ellipse in its class colour (later instances on top), a little noise on everything"""
    b = O.synthetic_batch(B, S, seed=seed)
    rng = np.random.RandomState(seed + 1000)
    img = 0.35 + 0.3 * b["images"].numpy()
    tb = b["true_boxes"].numpy().reshape(B, -1, 5)
    for i in range(B):
        for j in range(cfg.MAX_BOX_PER_IMAGE):
            m = b["true_masks"][i, j]
            if m.any():
                img[i][m] = CLASS_COLOUR[int(tb[i, j, 4])]
    img += rng.normal(0.0, 0.02, img.shape).astype(np.float32)
    b["images"] = torch.from_numpy(np.clip(img, 0.0, 1.0).astype(np.float32))
    return b


def train_overfit(dev, batch, B: int, S: int, steps: int, seed: int = 0, jitter: float = 0.02):
    """stage-1 recorded HIP training on one batch; returns (state_dict on the CPU, loss curve).

    ``jitter``: every step sees the batch's images plus a FRESH draw of N(0, jitter^2) pixel noise (made on the device).
    Without it the fit memorises the deterministic bf16 rounding pattern of the one forward pass it is trained through:
    round 5's per-pair table showed the HIP path's confidence logits at the trained batch size sitting +1 ... +3 above
    the f32 oracle's at EVERY detected cell of the 832^2 fixture, while the HIP path at another batch size (other
    tiles, other summation order) and the bf16-emulating oracle scattered around it by +-0.3 -- the heads had learnt
    the backbone's rounding noise on those four images.  The reference trains on freshly augmented images every step
    (utils/train_data.py:321-531); a fixture that does not is measuring its own memorisation."""
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=seed)
    net.set_batch(batch)
    net.shuffle_seed = 11
    net.build_program()
    clean = batch["images"].to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + seed)
    curve = []
    for t in range(steps):
        if jitter > 0:
            noise = torch.randn(clean.shape, generator=gen, device=dev, dtype=torch.float32)
            net.images[:B].copy_(torch.clamp(clean + jitter * noise, 0.0, 1.0))
        loss = net.train_step(None, want_loss=(t % max(steps // 6, 1) == 0 or t == steps - 1))
        if loss is not None:
            curve.append(float(loss.cpu()))
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    del net
    torch.cuda.empty_cache()
    return sd, curve


def box_iou(a, b) -> float:
    iy = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    ix = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iy * ix
    ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return float(inter / ua) if ua > 0 else 0.0


def match_image(ref_box, ref_mask, got_box, got_mask, iou_min: float = 0.9):
    """greedy one-to-one matching, oracle detections in score order: the unmatched HIP detection of the same class with
    the highest IoU >= iou_min.  Returns per-pair rows (box IoU, |score difference|, mask IoU after > 0.5) and the
    counts (oracle detections, reproduced, HIP detections, unmatched HIP)."""
    used = set()
    rows, missed = [], []
    for r in range(len(ref_box)):
        best, best_j = 0.0, -1
        for j in range(len(got_box)):
            if j in used or got_box[j][4] != ref_box[r][4]:
                continue
            v = box_iou(ref_box[r], got_box[j])
            if v > best:
                best, best_j = v, j
        if best_j >= 0 and best >= iou_min:
            used.add(best_j)
            a = np.asarray(ref_mask[r]) > 0.5
            g = np.asarray(got_mask[best_j]) > 0.5
            union = np.logical_or(a, g).sum()
            miou = float(np.logical_and(a, g).sum() / union) if union else 1.0
            rows.append((best, abs(float(ref_box[r][5]) - float(got_box[best_j][5])), miou))
        else:
            # (oracle score, its box size in pixels of the S/2 map, best same-class IoU the HIP path offers)
            size = np.round(np.asarray(ref_box[r][:4]) * np.asarray(ref_mask).shape[-1])
            missed.append((round(float(ref_box[r][5]), 4), [int(size[2] - size[0]), int(size[3] - size[1])], round(best, 4)))
    return rows, (len(ref_box), len(rows), len(got_box), len(got_box) - len(used)), missed


def flat_logits(y) -> torch.Tensor:
    """three head tensors [B,g,g,3,8] (or [B,g,g,24]), 72-, 36-, 18-grid order -> [B, candidates, 8] f32 in the candidate
    order of filter_detections (yolo/yolo3_net_pos.py:527-538): row = (tx, ty, tw, th, conf, class0..2) pre-sigmoid"""
    return torch.cat([t.detach().float().reshape(t.shape[0], -1, 8) for t in y], dim=1)


def candidate_score(row: np.ndarray, classid: int) -> float:
    """sigmoid(conf) * softmax(class)[classid] of one candidate's raw row (:541-548), f64"""
    r = np.asarray(row, np.float64)
    e = np.exp(r[5:] - r[5:].max())
    return float(1.0 / (1.0 + np.exp(-r[4])) * e[classid] / e.sum())


def _oracle_eval(sd, images, window, det_thresh, quant=None):
    """-> (boxes per image, masks per image, [3 logit tensors + score maps], det [B,30,6], candidate index [B,30])"""
    with torch.no_grad():
        y, m = O.build_network(sd, images, False, O.default_lock(1), quant=quant)
        pred = O.interpret_output(y)
        det, idx = O.filter_detections(pred[2], pred[3], pred[5], window, det_thresh, return_index=True)
        box, mask = O.val_test(det, m)
    return box, mask, [t.float() for t in y] + [m.float()], det, idx


KINDS = (("txy", slice(0, 2)), ("twh", slice(2, 4)), ("conf", slice(4, 5)), ("cls", slice(5, 8)))


def pair_table(ref_box, ref_det, ref_idx, flat_f32, flat_bf16, flat_hip, got_box, got_mask, ref_mask, S: int):
    """One row per detection of the f32 oracle: the raw (pre-sigmoid) head outputs AT THE CANDIDATE THE ORACLE DETECTED
    (same grid cell, same anchor) in the three forward passes -- f32 oracle, bf16-emulating oracle (the yardstick), HIP --
    and, when the HIP path's detections hold a same-class box with IoU >= 0.9, that pair's score difference.
    d_hip[k] / d_yard[k] = max over kind k's components of |HIP - f32| resp. |bf16 oracle - f32|."""
    rows = []
    empty = np.zeros((0, S // 2, S // 2), np.float32)
    for b in range(len(ref_box)):
        rb = ref_box[b]
        gb = got_box[b]
        rm = empty if np.ndim(ref_mask[b]) == 0 else ref_mask[b]
        gm = empty if np.ndim(got_mask[b]) == 0 else got_mask[b]
        used = set()
        for r in range(len(rb)):
            # candidate index of this row: its position in the filter's [30, 6] output (val_test may have dropped
            # zero-area rows in front of it)
            q = [j for j in range(ref_det.shape[1]) if ref_idx[b, j] >= 0 and np.array_equal(ref_det[b, j], rb[r])]
            assert q, "detection row not found in the filter's output"
            c = int(ref_idx[b, q[0]])
            cid = int(rb[r][4])
            f, y, h = flat_f32[b, c].numpy(), flat_bf16[b, c].numpy(), flat_hip[b, c].numpy()
            row = {"image": b, "candidate": c, "class": cid, "score_f32": float(rb[r][5]),
                   "score_bf16_same_cell": candidate_score(y, cid), "score_hip_same_cell": candidate_score(h, cid),
                   "raw_f32": [round(float(v), 5) for v in f], "raw_bf16": [round(float(v), 5) for v in y],
                   "raw_hip": [round(float(v), 5) for v in h]}
            for k, sl in KINDS:
                row["d_hip_" + k] = float(np.abs(h[sl].astype(np.float64) - f[sl]).max())
                row["d_yard_" + k] = float(np.abs(y[sl].astype(np.float64) - f[sl]).max())
            row["d_yard_score"] = abs(row["score_bf16_same_cell"] - row["score_f32"])
            row["d_hip_score_same_cell"] = abs(row["score_hip_same_cell"] - row["score_f32"])
            best, best_j = 0.0, -1
            for j in range(len(gb)):
                if j in used or gb[j][4] != rb[r][4]:
                    continue
                v = box_iou(rb[r], gb[j])
                if v > best:
                    best, best_j = v, j
            row["pair_box_iou"] = round(best, 4)
            if best_j >= 0 and best >= 0.9:
                used.add(best_j)
                row["pair_score_diff"] = abs(float(rb[r][5]) - float(gb[best_j][5]))
                a, g = np.asarray(rm[r]) > 0.5, np.asarray(gm[best_j]) > 0.5
                u = np.logical_or(a, g).sum()
                row["pair_mask_iou"] = float(np.logical_and(a, g).sum() / u) if u else 1.0
            else:
                row["pair_score_diff"] = None
                row["pair_mask_iou"] = None
            rows.append(row)
    return rows


# per-pair bars of tests/test_gpu_e2e_parity.py: |HIP - f32| <= YARD_FACTOR * |bf16 oracle - f32| + EPS[kind] at the
# same candidate.  EPS = the spread two bf16 implementations with different summation orders show against EACH OTHER at
# a detected cell (the bf16 oracle and the HIP path are two draws of the same rounding process, so one of them can sit
# on the f32 value by chance: the yardstick alone would be a coin flip): measured in profiles/r05_e2e_parity.json.
YARD_FACTOR = 1.5
EPS = {"txy": 0.15, "twh": 0.05, "conf": 0.25, "cls": 0.2, "score": 0.03}
SCORE_BAR = 0.12


def pair_violations(rows):
    """the rows (with the reason) that break a per-pair bar"""
    bad = []
    for r in rows:
        why = []
        for k, _ in KINDS:
            if r["d_hip_" + k] > YARD_FACTOR * r["d_yard_" + k] + EPS[k]:
                why.append("%s: |hip-f32| %.3f > %.1f * %.3f + %.2f" % (k, r["d_hip_" + k], YARD_FACTOR, r["d_yard_" + k], EPS[k]))
        sd = r["pair_score_diff"]
        if sd is not None and sd > SCORE_BAR and sd > YARD_FACTOR * r["d_yard_score"] + EPS["score"]:
            why.append("score: pair differs by %.3f > %.2f and > %.1f * %.3f + %.2f" % (sd, SCORE_BAR, YARD_FACTOR, r["d_yard_score"], EPS["score"]))
        if why:
            bad.append(dict(r, why=why))
    return bad


def _agreement(ref_box, ref_mask, got_box, got_mask, S: int, det_thresh: float, iou_min: float = 0.9):
    """matching statistics of ``got`` against the reference ``ref`` over a batch"""
    rows, missed, n_ref, n_rep, n_got, n_extra, loose = [], [], 0, 0, 0, 0, 0
    n_conf = conf_loose = 0
    empty = np.zeros((0, S // 2, S // 2), np.float32)
    for b in range(len(ref_box)):
        rb, rm = ref_box[b], (empty if np.ndim(ref_mask[b]) == 0 else ref_mask[b])
        gb, gm = got_box[b], (empty if np.ndim(got_mask[b]) == 0 else got_mask[b])
        r, (a, c, d, e), ms = match_image(rb, rm, gb, gm, iou_min)
        rows += r
        missed += ms
        n_ref += a
        n_rep += c
        n_got += d
        n_extra += e
        # the looser question: is the object found at all (same class, IoU >= 0.75)?
        _, (_, c75, _, _), _ = match_image(rb, rm, gb, gm, 0.75)
        loose += c75
        # ... among the reference detections that are not next to the score threshold
        conf = rb[:, 5] >= det_thresh + 0.1
        _, (a2, c2, _, _), _ = match_image(rb[conf], rm[conf], gb, gm, 0.75)
        n_conf += a2
        conf_loose += c2
    rows = np.asarray(rows, np.float64).reshape(-1, 3)
    scores = np.concatenate([rb[:, 5] for rb in ref_box]) if n_ref else np.zeros(0)
    return {
        "ref_detections": n_ref, "hip_detections": n_got,
        "reproduced_iou90": n_rep, "reproduced_iou90_frac": (n_rep / n_ref) if n_ref else None,
        "reproduced_iou75": loose, "reproduced_iou75_frac": (loose / n_ref) if n_ref else None,
        "confident_ref": n_conf, "confident_reproduced_iou75": conf_loose,
        "hip_unmatched": n_extra, "ref_missed": missed,
        "box_iou_min": float(rows[:, 0].min()) if len(rows) else None,
        "score_absdiff_max": float(rows[:, 1].max()) if len(rows) else None,
        "score_absdiff_2nd": float(np.sort(rows[:, 1])[-2]) if len(rows) > 1 else (float(rows[:, 1].max()) if len(rows) else None),
        "score_absdiff_median": float(np.median(rows[:, 1])) if len(rows) else None,
        "mask_iou_min": float(rows[:, 2].min()) if len(rows) else None,
        "mask_iou_mean": float(rows[:, 2].mean()) if len(rows) else None,
        "ref_score_min": float(scores.min()) if n_ref else None,
        "ref_score_median": float(np.median(scores)) if n_ref else None,
    }


def oracle_pair(sd, images, window, det_thresh):
    """the two oracle passes of one fixture (f32, bf16-emulating); inference is per image, so a caller comparing a
    sub-batch slices these instead of running the CPU network again"""
    return {"f32": _oracle_eval(sd, images, window, det_thresh),
            "bf16": _oracle_eval(sd, images, window, det_thresh, quant=O.bf16_ste)}


def _slice_eval(ev, n):
    box, mask, logits, det, idx = ev
    return box[:n], mask[:n], [t[:n] for t in logits], det[:n], idx[:n]


def compare(dev, sd, images: torch.Tensor, window: np.ndarray, S: int, det_thresh: float, net_kwargs=None,
            oracle=None):
    """both paths on the same images / variables.  Returns {"vs_f32": ..., "vs_bf16": ..., "bf16_vs_f32": ..., "pairs":
    ...}: the HIP path against the f32 oracle (the north_star's sentence), against the oracle that rounds every stored
    tensor to bf16 where the kernels do (same arithmetic, another summation order), -- the yardstick -- that
    bf16-emulating oracle against the f32 one: what bf16 storage itself costs, whoever implements it; and the per-pair
    table (pair_table).  ``oracle`` = oracle_pair() of a batch whose first images are ``images`` (else computed)."""
    B = images.shape[0]
    net = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0, **(net_kwargs or {}))
    net.load_state_dict(sd)
    got_box, got_mask = net.evaluation(images, window, [det_thresh])
    torch.cuda.synchronize()
    logits = [net.by_idx[i].act.float().cpu().clone() for i in (75, 67, 59, 82)]
    del net
    torch.cuda.empty_cache()

    def rel(xs, ws):
        return [round(float((g.double().reshape(-1) - w.double().reshape(-1)).norm() / w.double().norm()), 5) for g, w in zip(xs, ws)]

    if oracle is None:
        oracle = oracle_pair(sd, images, window, det_thresh)
    fb, fm, fl, fdet, fidx = _slice_eval(oracle["f32"], B)
    qb, qm, ql, _, _ = _slice_eval(oracle["bf16"], B)
    out = {"S": S, "B": B, "det_thresh": det_thresh,
           "vs_f32": dict(_agreement(fb, fm, got_box, got_mask, S, det_thresh), logit_rel_l2=rel(logits, fl)),
           "vs_bf16": dict(_agreement(qb, qm, got_box, got_mask, S, det_thresh), logit_rel_l2=rel(logits, ql)),
           "bf16_vs_f32": dict(_agreement(fb, fm, qb, qm, S, det_thresh), logit_rel_l2=rel(ql, fl))}
    out["pairs"] = pair_table(fb, fdet, fidx, flat_logits(fl[:3]), flat_logits(ql[:3]), flat_logits(logits[:3]),
                              got_box, got_mask, fm, S)
    out["pair_violations"] = pair_violations(out["pairs"])
    return out
