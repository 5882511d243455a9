"""Known-answer edge cases fed DIRECTLY to the HIP entry points (not through the network) and
compared with the oracle bit for bit: rounding at x.5, truncated bin edges, NMS ties and the strict
comparisons, zero-area RoIs (NaN), boxes straddling the clip window / the map.

Everything here is built from values that are exact in f32 on both sides (logits 0 / +-40, power-of
-two grids, dyadic box coordinates), so "equal" means np.array_equal -- no tolerance hides a flipped
comparison (yolo/yolo3_net_pos.py:558, 568-572, 810-813, 842, 876-878)."""
import numpy as np
import pytest
import torch

import disyolo_oracle as O
from disyolo_amd import lib as L
from disyolo_amd import config as cfg

pytestmark = pytest.mark.gpu

S = 64                                   # grids 8 / 4 / 2: cell centres and anchor/S are dyadic
ANCH = np.array([[16, 16], [32, 32], [8, 24],        # 8-grid
                 [32, 32], [16, 48], [48, 16],       # 4-grid
                 [32, 32], [64, 32], [32, 64]], np.float32)


def blank_logits(B):
    """conf logit -40 everywhere: sigmoid = 4e-18, far below any threshold"""
    ys = [torch.zeros(B, g, g, 3, 8) for g in (8, 4, 2)]
    for y in ys:
        y[..., 4] = -40.0
    return ys


def put(y, b, cy, cx, a, cls, conf=0.0):
    """a candidate with score sigmoid(conf) * 1.0 exactly (class margin 40 -> softmax max == 1.0f)"""
    y[b, cy, cx, a, :4] = 0.0           # sigmoid(0) = .5 -> centre of the cell; exp(0) = 1 -> anchor size
    y[b, cy, cx, a, 4] = conf
    y[b, cy, cx, a, 5:] = 0.0
    y[b, cy, cx, a, 5 + cls] = 40.0


def run_detect(dev, ys, window, thr, nms_thr, max_det=cfg.MAX_DETECTION):
    B = ys[0].shape[0]
    det = torch.full((B, max_det, 6), float("nan"), device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    logits = [y.reshape(B, y.shape[1], y.shape[2], 24).contiguous().to(dev) for y in ys]
    L.detect(logits[0], logits[1], logits[2], B, S, 3, ANCH.reshape(-1), torch.as_tensor(window, device=dev).float(),
             float(thr), float(nms_thr), max_det, det, cnt, L.Workspace(dev))
    torch.cuda.synchronize()
    pred = O.interpret_output(ys, anchors=ANCH)
    want = O.filter_detections(pred[2], pred[3], pred[5], np.asarray(window, np.float32), thr, nms_thr, max_det)
    return det.cpu().numpy(), cnt.cpu().numpy(), want


def test_detect_score_exactly_at_threshold_is_dropped(dev):
    ys = blank_logits(1)
    put(ys[1], 0, 1, 1, 0, cls=2)                     # score = 0.5 exactly
    win = [[0, 0, 1, 1]]
    got, cnt, want = run_detect(dev, ys, win, 0.5, 0.3)
    assert (want == 0).all() and cnt[0] == 0          # strict '>' (:558)
    np.testing.assert_array_equal(got, want)
    below = float(np.nextafter(np.float32(0.5), np.float32(0)))
    got, cnt, want = run_detect(dev, ys, win, below, 0.3)
    assert cnt[0] == 1 and want[0, 0, 5] == 0.5 and want[0, 0, 4] == 2
    np.testing.assert_array_equal(got, want)
    # box = cell centre (1.5/4) -/+ 32/64/2
    np.testing.assert_array_equal(got[0, 0, :4], np.float32([0.125, 0.125, 0.625, 0.625]))


def test_detect_iou_exactly_at_threshold_survives_and_just_above_is_suppressed(dev):
    # two 0.5 x 0.5 boxes, centres one 4-grid cell apart: inter .125, union .375 -> IoU = 1/3 in f32
    ys = blank_logits(1)
    put(ys[1], 0, 1, 1, 0, cls=1, conf=2.0)
    put(ys[1], 0, 1, 2, 0, cls=1, conf=1.0)
    win = [[0, 0, 1, 1]]
    third = float(np.float32(0.125) / np.float32(0.375))
    got, cnt, want = run_detect(dev, ys, win, 0.25, third)
    assert cnt[0] == 2                                # IoU > thr is false at equality (:568-572)
    np.testing.assert_array_equal(got, want)
    got, cnt, want = run_detect(dev, ys, win, 0.25, float(np.nextafter(np.float32(third), np.float32(0))))
    assert cnt[0] == 1 and (want[0, 0, 5] > 0.8)
    np.testing.assert_array_equal(got, want)
    # a different class is never suppressed by it
    put(ys[1], 0, 1, 2, 0, cls=0, conf=1.0)
    got, cnt, want = run_detect(dev, ys, win, 0.25, 0.1)
    assert cnt[0] == 2
    np.testing.assert_array_equal(got, want)


def test_detect_equal_scores_keep_candidate_order(dev):
    """ties: the lower flattened index (scale 8-grid first, then y, x, anchor) is visited first by the
    NMS and listed first by top_k"""
    ys = blank_logits(2)
    # image 0: two overlapping equal-score boxes of one class -> the first in candidate order survives
    put(ys[1], 0, 2, 1, 0, cls=0, conf=1.0)
    put(ys[1], 0, 2, 2, 0, cls=0, conf=1.0)
    # image 1: four equal-score boxes far apart, on three scales and two classes
    put(ys[2], 1, 0, 0, 0, cls=1, conf=1.0)
    put(ys[0], 1, 6, 6, 0, cls=2, conf=1.0)
    put(ys[0], 1, 1, 6, 0, cls=1, conf=1.0)
    put(ys[1], 1, 3, 0, 0, cls=2, conf=1.0)
    win = [[0, 0, 1, 1], [0, 0, 1, 1]]
    got, cnt, want = run_detect(dev, ys, win, 0.25, 0.2)
    assert list(cnt) == [1, 4]
    np.testing.assert_array_equal(got, want)
    assert want[0, 0, 3] == np.float32(0.625)          # the x = 1 cell's box won
    # image 1 order = candidate index order: 8-grid (1,6), 8-grid (6,6), 4-grid, 2-grid
    np.testing.assert_array_equal(want[1, :4, 4], [1, 2, 2, 1])


def test_detect_clips_to_the_window_before_nms_and_caps_at_max_detection(dev):
    ys = blank_logits(2)
    # image 0: 8-grid fully populated with anchor 0 (16 px boxes, disjoint), alternating scores ->
    # 64 candidates > 30: top-30 by score, ties by index
    for cy in range(8):
        for cx in range(8):
            put(ys[0], 0, cy, cx, 0, cls=(cy + cx) % 3, conf=float((cy * 8 + cx) % 5))
    # image 1: boxes straddling a window; clipping changes their IoU (two boxes become identical strips)
    put(ys[1], 1, 0, 1, 0, cls=0, conf=2.0)
    put(ys[1], 1, 0, 2, 0, cls=0, conf=1.0)
    put(ys[2], 1, 1, 1, 1, cls=1, conf=3.0)            # 64 x 32 anchor: wider than the image
    win = [[0, 0, 1, 1], [0.25, 0.25, 0.75, 0.75]]
    got, cnt, want = run_detect(dev, ys, win, 0.25, 0.3)
    assert cnt[0] == 30 and cnt[1] >= 2
    np.testing.assert_array_equal(got, want)
    assert want[1, :cnt[1], :4].min() >= 0.25 and want[1, :cnt[1], :4].max() <= 0.75


@pytest.mark.parametrize("case", ["balanced", "one_class", "sparse", "ties"])
def test_detect_greedy_nms_on_thousands_of_candidates(dev, case):
    """The detection filter's greedy loop at the sizes an untrained network produces (576^2: 20,412 candidates per image,
    thousands per class over the threshold) -- list in registers (<= 8,192 per class), list in global memory (one class
    takes more), a handful of candidates, and blocks of exactly equal scores -- replayed by the oracle's
    non_max_suppression + top-k merge on the boxes / scores / classes the decode kernel itself produced (read back from
    the workspace), so the comparison is exact: any difference is a different NMS decision."""
    S_, B = 576, 2
    g = torch.Generator().manual_seed({"balanced": 1, "one_class": 2, "sparse": 3, "ties": 4}[case])
    ys = [torch.randn(B, gs, gs, 3, 8, generator=g) for gs in (72, 36, 18)]
    thr = 0.25
    for y in ys:
        y[..., 2:4] *= 0.5
        if case == "one_class":
            y[..., 4] += 3.0
            y[..., 5] += 4.0                      # nearly every candidate is class 0 and passes: > 8,192 in one list
        elif case == "sparse":
            y[..., 4] -= 3.0
        elif case == "ties":
            y[..., 4] = torch.round(y[..., 4])               # few distinct scores: long runs of equal ones
            y[..., 5:] = torch.round(y[..., 5:]) * 40.0
        else:
            y[..., 4] += 1.5
    anchors = np.asarray(cfg.ANCHORS, np.float32).reshape(-1)
    win = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.1, 0.05, 0.9, 0.95]], device=dev)
    max_det = cfg.MAX_DETECTION
    nms_f32 = float(np.float32(cfg.IOU_THRESHOLD))     # the kernel compares in f32
    det = torch.full((B, max_det, 6), float("nan"), device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    logits = [y.reshape(B, y.shape[1], y.shape[2], 24).contiguous().to(dev) for y in ys]
    ws = L.Workspace(dev)
    L.detect(logits[0], logits[1], logits[2], B, S_, 3, anchors, win, thr, cfg.IOU_THRESHOLD, max_det, det, cnt, ws)
    torch.cuda.synchronize()
    NC = 3 * (72 * 72 + 36 * 36 + 18 * 18)
    raw = ws.buf.cpu().numpy()
    boxes = raw[:B * NC * 16].view(np.float32).reshape(B, NC, 4)
    scores = raw[B * NC * 16:B * NC * 20].view(np.float32).reshape(B, NC)
    classes = raw[B * NC * 20:B * NC * 24].view(np.int32).reshape(B, NC)
    want = np.zeros((B, max_det, 6), np.float32)
    biggest = 0
    for b in range(B):
        keep = np.where(scores[b] > np.float32(thr))[0]
        kept = []
        for c in np.unique(classes[b][keep]):
            ixs = keep[classes[b][keep] == c]
            biggest = max(biggest, len(ixs))
            # (the oracle's loop, candidates pre-sorted with numpy: its pure-Python sort key is the slow part at this size)
            order = np.lexsort((np.arange(len(ixs)), -scores[b][ixs].astype(np.float64)))
            sel = []
            for j in order:
                if len(sel) >= max_det:
                    break
                if all(O._tf_iou(boxes[b][ixs[j]], boxes[b][ixs[q]]) <= nms_f32 for q in sel):
                    sel.append(j)
            kept.extend(int(ixs[q]) for q in sel)
        kept = np.array(sorted(set(kept)), dtype=np.int64)
        order = sorted(range(len(kept)), key=lambda q: (-float(scores[b][kept[q]]), q))[:max_det]
        for r, q in enumerate(order):
            want[b, r, :4] = boxes[b][kept[q]]
            want[b, r, 4] = classes[b][kept[q]]
            want[b, r, 5] = scores[b][kept[q]]
        assert int(cnt[b]) == len(order)
    if case == "one_class":
        assert biggest > 8192
    elif case in ("balanced", "ties"):
        assert 1024 < biggest <= 8192
    else:
        assert 0 < biggest < 1024
    np.testing.assert_array_equal(det.cpu().numpy(), want)


# ---------------------------------------------------------------------------------------------
SM = 32          # score-map size (S/2): k/32 is dyadic, so box*SM lands exactly on x.5


def run_assemble(dev, det, score):
    B = det.shape[0]
    masks = torch.full((B, det.shape[1], SM, SM), float("nan"), device=dev)
    keep = torch.zeros(B, det.shape[1], dtype=torch.int32, device=dev)
    L.psroi_assemble(score.to(dev), torch.as_tensor(det, device=dev), B, det.shape[1], SM, 3, masks, keep)
    torch.cuda.synchronize()
    return masks.cpu().numpy(), keep.cpu().numpy().astype(bool)


def coded_score(B, seed=0):
    """values from a coarse table: neighbouring table entries differ by >= 0.01 after the sigmoid, so a
    tolerance of 1e-6 on the output proves WHICH channel / pixel was selected"""
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(-8, 9, (B, SM, SM, 9), generator=g).float() * 0.5).contiguous()


def test_assemble_rounds_half_to_even_and_drops_sub_pixel_boxes(dev):
    det = np.zeros((1, 30, 6), np.float32)
    # y1 = 2.5 -> 2, x1 = 3.5 -> 4, y2 = 20.5 -> 20, x2 = 21.5 -> 22   (tf.round, :876)
    det[0, 0] = [2.5 / 32, 3.5 / 32, 20.5 / 32, 21.5 / 32, 1, 0.9]
    det[0, 1] = [0.3 / 32, 0.1, 0.45 / 32, 0.6, 0, 0.8]      # rounds to zero height -> dropped (:877-878)
    det[0, 2] = [0.2, 10.5 / 32, 0.7, 10.5 / 32, 2, 0.7]     # zero width
    det[0, 3] = [0.5 / 32, 0.5 / 32, 1.5 / 32, 1.5 / 32, 0, 0.6]   # 0.5 -> 0, 1.5 -> 2: a 2x2 box survives
    det[0, 5] = [0.25, 0.25, 0.75, 0.75, 0, 0.5]             # after a zero row (padding in the middle)
    score = coded_score(1)
    masks, keep = run_assemble(dev, det, score)
    wb, wm = O.val_test(det, score)
    assert list(np.where(keep[0])[0]) == [0, 3, 5]
    np.testing.assert_array_equal(det[0][keep[0]], wb[0])
    np.testing.assert_allclose(masks[0][keep[0]], wm[0], rtol=0, atol=1e-6)
    m0 = masks[0, 0]
    assert (m0[:2] == 0.5).all() and (m0[20:] == 0.5).all() and (m0[:, :4] == 0.5).all() and (m0[:, 22:] == 0.5).all()
    assert (m0[2:20, 4:22] != 0.5).any()
    # bin edges of rows 2..20: int(2), round(8), round(14), int(20)
    assert O.kmask_edges(2, 20) == [2, 8, 14, 20] and O.kmask_edges(4, 22) == [4, 10, 16, 22]


def test_assemble_bin_edges_truncate_and_round_on_thirds(dev):
    # widths that are not multiples of 3: lo + w/3 lands on x.333 / x.667 and (7-0)/3*... on x.5 never for
    # integers, so every edge is decided by rintf/int alone
    det = np.zeros((2, 30, 6), np.float32)
    boxes = [(0, 0, 7, 10), (3, 5, 31, 32), (1, 2, 2, 3), (0, 0, 32, 32), (10, 9, 21, 29), (30, 30, 32, 32)]
    for r, (y1, x1, y2, x2) in enumerate(boxes):
        det[r % 2, r // 2] = [y1 / 32, x1 / 32, y2 / 32, x2 / 32, r % 3, 0.9 - 0.1 * r]
    score = coded_score(2, seed=1)
    masks, keep = run_assemble(dev, det, score)
    wb, wm = O.val_test(det, score)
    for b in range(2):
        assert keep[b].sum() == 3
        np.testing.assert_allclose(masks[b][keep[b]], wm[b], rtol=0, atol=1e-6)


def test_assemble_without_detections_keeps_nothing(dev):
    det = np.zeros((1, 30, 6), np.float32)
    masks, keep = run_assemble(dev, det, coded_score(1))
    wb, wm = O.val_test(det, coded_score(1))
    assert not keep.any() and np.ndim(wm[0]) == 0 and wm[0] == 0.0       # scalar 0.0 (:933)


def run_mask_loss(dev, det, tb, tm, score, perms=None):
    B = det.shape[0]
    G = cfg.MAX_BOX_PER_IMAGE
    rois = torch.zeros(B, L.ROI_MAX, L.ROI_W, dtype=torch.int32, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    pd = torch.arange(30, dtype=torch.int32, device=dev).repeat(B, 1).contiguous()
    pg = torch.arange(G, dtype=torch.int32, device=dev).repeat(B, 1).contiguous()
    if perms is not None:
        pd = torch.as_tensor(np.stack([p[0] for p in perms]), device=dev).int().contiguous()
        pg = torch.as_tensor(np.stack([p[1] for p in perms]), device=dev).int().contiguous()
    L.mask_rois(torch.as_tensor(det, device=dev), 30, torch.as_tensor(tb.reshape(B, G, 5), device=dev), G, pd, pg, B, SM,
                cfg.MASK_ROI_DET, cfg.MASK_ROI_GT, cfg.MASK_ROI_IOU, rois, cnt)
    dscore = torch.zeros(B, SM, SM, L.GRAD_LD, dtype=torch.bfloat16, device=dev)
    loss = torch.zeros(1, device=dev)
    L.psroi_loss(score.to(dev), torch.as_tensor(tm, device=dev).to(torch.uint8).contiguous(), G, rois, cnt, B, SM, 3,
                 cfg.MASK_SCALE, dscore, loss, L.Workspace(dev))
    torch.cuda.synchronize()
    return rois.cpu().numpy(), cnt.cpu().numpy(), float(loss.cpu()[0]), dscore.float().cpu()


def expected_rois(det, tb, perms=None):
    """the RoI table the kernel must produce, from the oracle's selection + bin edges"""
    out = []
    for i in range(det.shape[0]):
        pd, pg = perms[i] if perms is not None else (None, None)
        pos, assign, gt_rows = O.select_mask_rois(det[i], tb[i, 0, 0, 0], pd, pg)
        rows = []
        for r in range(len(pos)):
            px = np.round(pos[r] * np.float32(SM))
            area = int((O.channel_index_map(px, SM) >= 0).sum())
            rows.append(O.kmask_edges(px[0], px[2]) + O.kmask_edges(px[1], px[3]) + [int(gt_rows[assign[r]]), area, 1, 0])
        out.append(rows)
    return out


def test_mask_rois_half_to_even_iou_threshold_and_map_straddling_boxes(dev):
    B, G = 2, cfg.MAX_BOX_PER_IMAGE
    tb = np.zeros((B, 1, 1, 1, G, 5), np.float32)
    tm = np.zeros((B, G, 2 * SM, 2 * SM), bool)
    det = np.zeros((B, 30, 6), np.float32)
    # image 0, GT 0: box 8.5..24.5 (y) x 4.5..28.5 (x) on the map -> tf.round goes to even: 8, 24, 4, 28
    tb[0, 0, 0, 0, 0] = [16.5 / 32, 16.5 / 32, 24 / 32, 16 / 32, 1]
    tm[0, 0, 17:49, 9:57] = True
    # GT 2 (row 1 is empty: trimmed list index != row): partly outside the image, x 24..40 on the map
    tb[0, 0, 0, 0, 2] = [1.0, 0.25, 0.5, 0.25, 0]
    tm[0, 2, 8:24, 48:64] = True
    # detection 0 = GT 0 shifted by a third of its width: inter .25, union .5 -> IoU == 0.5 exactly in f32
    # -> positive (>=, :787); its right edge 36.5 -> 36 lies beyond the map
    det[0, 0] = [8.5 / 32, 12.5 / 32, 24.5 / 32, 36.5 / 32, 1, 0.9]
    det[0, 1] = [8.5 / 32, 4.5 / 32, 24.5 / 32, 28.5 / 32, 1, 0.8]              # IoU 1
    det[0, 2] = [0.0, 0.0, 0.2, 0.2, 0, 0.7]                                    # IoU 0 -> negative
    # shrink detection 0's overlap by one ulp-ish step -> just below 0.5 -> negative
    det[0, 3] = [8.5 / 32, 12.75 / 32, 24.5 / 32, 36.75 / 32, 1, 0.6]
    # image 1: a GT whose box rounds to zero area on the map but has IoU 1 with itself -> 0/0 = NaN (B14)
    tb[1, 0, 0, 0, 0] = [0.5, 0.5, 0.01, 0.01, 2]
    tm[1, 0, 31:33, 31:33] = True
    score = coded_score(B, seed=2)
    want = expected_rois(det, tb)
    rois, cnt, loss, dscore = run_mask_loss(dev, det, tb, tm, score)
    assert [len(w) for w in want] == list(cnt) and cnt[0] == 4 and cnt[1] == 1     # det 0, det 1, GT 0, GT 2
    for b in range(B):
        np.testing.assert_array_equal(rois[b, :cnt[b]], np.array(want[b], np.int32).reshape(cnt[b], L.ROI_W))
        assert (rois[b, cnt[b]:] == 0).all()
    assert want[0][0][:8] == [8, 13, 19, 24, 12, 20, 28, 36] and want[0][1][:8] == [8, 13, 19, 24, 4, 12, 20, 28]
    assert want[0][3][4:8] == [24, 29, 35, 40]                    # GT 2 straddles the right edge of the map
    assert want[1][0][9] == 0                                     # the zero-area positive RoI
    lm = O.loss_mask(det, score, tb, tm)
    assert np.isnan(float(lm)) and np.isnan(loss)                 # NaN propagates into the batch mean like TF
    # image 0 alone is finite and matches
    rois0, cnt0, loss0, ds0 = run_mask_loss(dev, det[:1], tb[:1], tm[:1], score[:1])
    sc = score[:1].clone().requires_grad_(True)
    lm0 = O.loss_mask(det[:1], sc, tb[:1], tm[:1])
    lm0.backward()
    np.testing.assert_allclose(loss0, float(lm0), rtol=2e-5)
    assert float(ds0[..., 9:].abs().max()) == 0.0
    try:
        np.testing.assert_allclose(ds0[..., :9].numpy(), sc.grad.numpy(), rtol=2 ** -7, atol=1e-9)
    except AssertionError:
        import os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        np.savez_compressed(os.path.join(out, "kat_dscore.npz"), got=ds0.numpy(), want=sc.grad.numpy(), rois=rois0, score=score.numpy())
        raise


def test_mask_rois_follow_the_injected_shuffle(dev):
    """first 7 of the shuffled proposals + first 3 of the shuffled GT boxes (:781-783)"""
    rng = np.random.RandomState(5)
    B, G = 3, cfg.MAX_BOX_PER_IMAGE
    b = O.synthetic_batch(B, 2 * SM, seed=17)
    tb = b["true_boxes"].numpy()
    det = np.zeros((B, 30, 6), np.float32)
    for i in range(B):
        rows = [r for r in range(G) if np.abs(tb[i, 0, 0, 0, r, :4]).sum() > 0]
        for q in range(12):
            xc, yc, w, h = tb[i, 0, 0, 0, rows[q % len(rows)], :4]
            j = (rng.rand(4) - 0.5) * 0.08
            det[i, q] = [yc - h / 2 + j[0], xc - w / 2 + j[1], yc + h / 2 + j[2], xc + w / 2 + j[3], q % 3, 0.9 - 0.01 * q]
    perms = [(rng.permutation(30).astype(np.int32), rng.permutation(G).astype(np.int32)) for _ in range(B)]
    score = coded_score(B, seed=3)
    want = expected_rois(det, tb, perms)
    rois, cnt, loss, _ = run_mask_loss(dev, det, tb, b["true_masks"], score, perms)
    assert list(cnt) == [len(w) for w in want] and sum(cnt) >= 6
    for i in range(B):
        np.testing.assert_array_equal(rois[i, :cnt[i]], np.array(want[i], np.int32).reshape(cnt[i], L.ROI_W))
    lm = O.loss_mask(det, score, tb, b["true_masks"], perms)
    np.testing.assert_allclose(loss, float(lm), rtol=2e-5)
