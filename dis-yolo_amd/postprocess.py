"""Caller side of ``YOLONet.evaluation``: the per-image post-processing of the reference's
``evaluate`` loop (calculate_test_map.py:203-269) -- letter-box window, box un-letterboxing, mask
crop / bilinear resize / threshold / paste -- with the pixel work on the GPU (disyolo_mask_paste).
The reference does it on the host with cv2 at ~0.1 s per image; at thousands of images per second
it has to sit next to the network.  SURVEY.md 8(f3)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from . import lib as L


def letterbox_window(image_h: int, image_w: int, size: int) -> np.ndarray:
    """``image_read``'s clip window (calculate_test_map.py:151-169): [top, left, bottom, right],
    normalised, of the aspect-preserving resize centred in the size x size letter box."""
    imgh, imgw = image_h, image_w
    if (float(size) / imgw) < (float(size) / imgh):
        imgh = (imgh * size) // imgw
        imgw = size
    else:
        imgw = (imgw * size) // imgh
        imgh = size
    top, left = (size - imgh) // 2, (size - imgw) // 2
    return np.array([top / size, left / size, (imgh + top) / size, (imgw + left) / size], np.float32)


def correct_yolo_boxes(boxes_yxyx: np.ndarray, image_h: int, image_w: int, net_h: int, net_w: int) -> np.ndarray:
    """calculate_test_map.py:121-138 for an [n,4] array of normalised (y1,x1,y2,x2) boxes: integer
    (x1,y1,x2,y2) pixel corners in the original image (half-to-even rounding, clamped)."""
    b = np.asarray(boxes_yxyx, np.float32).reshape(-1, 4)
    if (float(net_w) / image_w) < (float(net_h) / image_h):
        new_w, new_h = net_w, (image_h * net_w) // image_w
    else:
        new_h, new_w = net_h, (image_w * net_h) // image_h
    x_off, x_scale = float((net_w - new_w) // 2) / net_w, float(new_w) / net_w
    y_off, y_scale = float((net_h - new_h) // 2) / net_h, float(new_h) / net_h

    def corner(v, off, scale, n):
        return np.clip(np.around((v - off) / scale * n).astype(np.int32), 0, n)
    return np.stack([corner(b[:, 1], x_off, x_scale, image_w), corner(b[:, 0], y_off, y_scale, image_h),
                     corner(b[:, 3], x_off, x_scale, image_w), corner(b[:, 2], y_off, y_scale, image_h)], axis=1)


def paste_detections(det_box, det_mask, image_h: int, image_w: int, net_size: int,
                     want_full: bool = True) -> Tuple[List[Dict], torch.Tensor]:
    """One image of ``evaluation``'s result -> (entries, merged) like the reference loop body
    (calculate_test_map.py:220-266): ``entries`` = [{index, classid, score, mask (bool CUDA tensor
    [H,W])}] for the detections with a non-empty box, ``merged`` = uint8 CUDA tensor [H,W] with
    class id + 1.  ``det_box`` [n,6] (numpy or tensor), ``det_mask`` CUDA f32 [n,S,S] or the
    scalar 0.0 the reference returns for an image without detections."""
    dev = det_mask.device if torch.is_tensor(det_mask) else torch.device("cuda", torch.cuda.current_device())
    merged = torch.zeros(image_h, image_w, dtype=torch.uint8, device=dev)
    box = det_box.detach().cpu().numpy() if torch.is_tensor(det_box) else np.asarray(det_box)
    if not torch.is_tensor(det_mask) or box.size == 0:
        return [], merged
    box = box.reshape(-1, 6).astype(np.float32)
    n, size = box.shape[0], int(det_mask.shape[-1])
    dst = correct_yolo_boxes(box[:, :4], image_h, image_w, net_size, net_size)          # x1,y1,x2,y2
    crop = np.around(box[:, :4] * np.float32(size)).astype(np.int32)                       # y1,x1,y2,x2 on the map
    rects = np.concatenate([crop, dst[:, [1, 0, 3, 2]]], axis=1).astype(np.int32)          # cy1,cx1,cy2,cx2,y1,x1,y2,x2
    ok = ((dst[:, 3] - dst[:, 1]) * (dst[:, 2] - dst[:, 0]) > 0) & (crop[:, 2] > crop[:, 0]) & (crop[:, 3] > crop[:, 1])
    rects[~ok] = 0
    rects_d = torch.from_numpy(rects).to(dev)
    cls_d = torch.from_numpy(box[:, 4].astype(np.int32)).to(dev)
    full = torch.empty(n, image_h, image_w, dtype=torch.uint8, device=dev) if want_full else None
    L.mask_paste(det_mask.contiguous(), rects_d, cls_d, image_h, image_w, full, merged)
    entries = []
    for k in range(n):
        if ok[k]:
            entries.append({"index": k, "classid": int(box[k, 4]), "score": float(box[k, 5]),
                            "mask": full[k].bool() if want_full else None})
    return entries, merged


class SegmentationAccuracy:
    """the mIoU block of ``evaluate`` (calculate_test_map.py:303-346): accumulate the pixel confusion
    counts of (ground-truth class map, merged detection map) pairs on the GPU, image by image."""

    def __init__(self, device):
        self.conf = torch.zeros(16, dtype=torch.int64, device=device)

    def add(self, true_map, pred_map: torch.Tensor) -> None:
        t = torch.as_tensor(true_map).to(self.conf.device, torch.uint8).contiguous()
        p = pred_map.to(self.conf.device, torch.uint8).contiguous()
        if t.shape != p.shape:
            raise ValueError("class maps differ in shape: %s vs %s" % (tuple(t.shape), tuple(p.shape)))
        L.confusion16(t, p, self.conf)

    def result(self) -> List[float]:
        """[bg_iou, crack_iou, spall_iou, rebar_iou, miou]"""
        c = self.conf.cpu().numpy().reshape(4, 4).astype(np.float64)
        ious = [c[k, k] / (c[k, :].sum() + c[:, k].sum() - c[k, k]) for k in range(4)]
        return ious + [float(np.mean(ious))]
