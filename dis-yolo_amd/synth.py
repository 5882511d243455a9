"""Seeded synthetic training batches of the reference's feed_dict shape.

Restates the two pieces of the reference's CPU data path that define the tensors the hot
path consumes: the batch layout of ``defect_train.get`` (utils/train_data.py:44-52,258-265)
and the anchor/cell target assignment (utils/train_data.py:149-178).  Images are uniform
noise and instances are axis-aligned ellipses -- there is no dataset in the reference tree.
"""
from __future__ import annotations

import numpy as np

from . import config as cfg


def assign_target_entries(boxes_px, cls, S: int, num_class: int):
    """utils/train_data.py:149-178 as a sparse list: best-IoU anchor of the 9 (boxes centred at the origin),
    cell = int(centre * grid / net); an occupied (cell, anchor) keeps its first box.  Returns, per scale (yolo3, yolo2,
    yolo1), {(yi, xi, anchor): row float32 [5 + C]} in assignment order -- the non-zero rows of the dense target grids."""
    g1 = S // 32
    sizes = [4 * g1, 2 * g1, g1]
    out = [dict() for _ in sizes]
    amax = np.asarray(cfg.ANCHORS, np.float32) / 2.0
    a_area = amax[:, 0] * amax[:, 1] * 4
    for b, c in zip(boxes_px, cls):
        half = np.asarray(b[2:4], np.float32) / 2.0
        inter = np.maximum(2 * np.minimum(half[None, :], amax), 0.0)
        ia = inter[:, 0] * inter[:, 1]
        iou = ia / (half[0] * half[1] * 4 + a_area - ia)
        if iou.max() <= 0:
            continue
        idx = int(np.argmax(iou))
        g = sizes[idx // 3]
        xi, yi = int(b[0] * g / S), int(b[1] * g / S)
        key = (yi, xi, idx % 3)
        if key in out[idx // 3]:
            continue
        row = np.zeros(5 + num_class, np.float32)
        row[0:4] = b[:4]
        row[4] = 1
        row[5 + int(c)] = 1.0
        out[idx // 3][key] = row
    return out


def assign_targets(boxes_px, cls, S: int, num_class: int):
    """the dense grids of utils/train_data.py:149-178: [g, g, 3, 5 + C] per scale (yolo3, yolo2, yolo1)"""
    g1 = S // 32
    yolos = [np.zeros((m * g1, m * g1, 3, 5 + num_class), np.float32) for m in (4, 2, 1)]
    for y, ent in zip(yolos, assign_target_entries(boxes_px, cls, S, num_class)):
        for (yi, xi, a), row in ent.items():
            y[yi, xi, a] = row
    return yolos


def synthetic_batch(B: int, S: int, seed: int = 1234, num_class: int = 3):
    """dict of numpy arrays: images f32 [B,S,S,3] in [0,1), clip_window [B,4] = (0,0,1,1),
    true_boxes [B,1,1,1,20,5] (xc,yc,w,h normalised, class), true_masks bool [B,20,S,S],
    yolo1/2/3 targets [B,g,g,3,5+C] with normalised boxes."""
    rng = np.random.RandomState(seed)
    G = cfg.MAX_BOX_PER_IMAGE
    images = rng.rand(B, S, S, 3).astype(np.float32)
    true_boxes = np.zeros((B, 1, 1, 1, G, 5), np.float32)
    true_masks = np.zeros((B, G, S, S), bool)
    ys = [np.zeros((B, S // d, S // d, 3, 5 + num_class), np.float32) for d in (8, 16, 32)]
    yy, xx = np.mgrid[0:S, 0:S]
    for b in range(B):
        bx, cl = [], []
        for j in range(rng.randint(1, 6)):
            w, h = rng.uniform(0.05, 0.6, size=2) * S
            cx, cy = rng.uniform(w / 2, S - w / 2), rng.uniform(h / 2, S - h / 2)
            m = ((xx - cx) / (w / 2)) ** 2 + ((yy - cy) / (h / 2)) ** 2 <= 1.0
            if not m.any():
                continue
            rows, cols = np.where(m)
            x1, x2, y1, y2 = cols.min(), cols.max(), rows.min(), rows.max()
            if x2 <= x1 or y2 <= y1:
                continue
            c = rng.randint(0, num_class)
            box = [(x1 + x2) / 2.0, (y1 + y2) / 2.0, float(x2 - x1), float(y2 - y1)]
            true_masks[b, j] = m
            true_boxes[b, 0, 0, 0, j, :4] = np.asarray(box, np.float32) / S
            true_boxes[b, 0, 0, 0, j, 4] = c
            bx.append(box)
            cl.append(c)
        for dst, t in zip(ys, assign_targets(np.asarray(bx, np.float32).reshape(-1, 4), cl, S, num_class)):
            t[..., 0:4] /= S
            dst[b] = t
    return {"images": images, "clip_window": np.tile(np.array([[0, 0, 1, 1]], np.float32), (B, 1)),
            "true_boxes": true_boxes, "true_masks": true_masks, "yolo3": ys[0], "yolo2": ys[1], "yolo1": ys[2]}
