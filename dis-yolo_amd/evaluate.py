"""Test / validation driver: the counterpart of ``calculate_test_map.py`` (``image_read`` :149-176,
``evaluate`` :180-347, class ``MAP``) and of ``utils/validation_map.py`` (``MAP.do_python_eval``
:104-198), on top of ``YOLONet.evaluation``.

Same flow as the reference -- letter box -> ``sess.run(net.evaluation)`` -> un-letterbox boxes, crop /
resize / threshold / paste masks -> per-class mask AP at IoU 0.5 (``voc_eval``) -> 4x4 pixel confusion ->
mIoU -- with the pixel work on the GPU (``disyolo_letterbox``, ``disyolo_mask_paste``,
``disyolo_confusion16``) instead of cv2 on the host.  Image files are decoded with PIL (host).
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import config as cfg
from . import lib as L
from .postprocess import SegmentationAccuracy, correct_yolo_boxes, paste_detections
from .voc_eval import voc_eval


def image_read(image_rgb, image_size: int, device=None, out: Optional[torch.Tensor] = None):
    """calculate_test_map.py:149-176: RGB uint8 [H,W,3] (numpy or CUDA tensor) -> (letter-boxed image f32
    CUDA [S,S,3] in [0,1], clip window f32 [4] = top, left, bottom, right)."""
    if not torch.is_tensor(image_rgb):
        image_rgb = torch.from_numpy(np.ascontiguousarray(image_rgb))
    if device is None:
        device = image_rgb.device if image_rgb.is_cuda else torch.device("cuda", torch.cuda.current_device())
    rgb = image_rgb.to(device, torch.uint8).contiguous()
    if out is None:
        out = torch.empty(image_size, image_size, 3, dtype=torch.float32, device=device)
    window = L.letterbox(rgb, out, image_size)
    return out, window


def load_image_rgb(path: str) -> np.ndarray:
    """cv2.cvtColor(cv2.imread(path), COLOR_BGR2RGB) (calculate_test_map.py:208) via PIL"""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


class MAP(object):
    """Ground truth + metric, like the reference's two ``MAP`` classes.  ``groundtruth`` =
    [recs_mask, recs_mergemask or recs_size, ..., index] is passed in (the reference builds it from its
    pickle cache with skimage; ``train_data.rasterize_polygons`` is the counterpart here):
      recs_mask: image id -> [{'imageid', 'classid', 'difficult', 'mask' bool [H,W]}]
      sizes:     image id -> [image_h, image_w]
      merged:    image id -> uint8 [H,W] class map (0 background, classid + 1), for mIoU; optional
      index:     list of image ids in evaluation order."""

    def __init__(self, recs_mask: Dict[str, List[Dict]], sizes: Dict[str, Sequence[int]], index: Sequence[str],
                 merged: Optional[Dict[str, np.ndarray]] = None, net_size: int = cfg.TEST_SIZE):
        self.num_class = len(cfg.CLASSES)
        self.classid = list(range(self.num_class))
        self.class_to_ind = dict(zip(cfg.CLASSES, range(self.num_class)))
        self.recs_mask, self.sizes, self.index, self.merged = recs_mask, sizes, list(index), merged
        self.net_size = net_size
        self.groundtruth = [recs_mask, merged, sizes, self.index]
        self._gt_cache = {}

    @staticmethod
    def correct_yolo_boxes(x1, y1, x2, y2, image_h, image_w, net_h, net_w):
        """calculate_test_map.py:121-138 (one box; returns x1, y1, x2, y2 integer pixel corners)"""
        b = correct_yolo_boxes(np.array([[y1, x1, y2, x2]], np.float32), image_h, image_w, net_h, net_w)[0]
        return int(b[0]), int(b[1]), int(b[2]), int(b[3])

    def _ap_table(self, detfile: Dict[str, List[Dict]], thresh: float = 0.5):
        """calculate_test_map.py:275-299 / validation_map.py:170-197"""
        res, pres, aps = [], [], []
        for clsid in self.classid:
            if not detfile[str(clsid)]:
                res, pres, aps = res + [0.0], pres + [0.0], aps + [0.0]
                continue
            recall, precision, ap = voc_eval(detfile[str(clsid)], self.recs_mask, self.index, clsid, ovthresh=thresh,
                                             use_07_metric=False)
            res, pres, aps = res + [recall], pres + [precision], aps + [ap]
        return [{"thresh": thresh, "AP": aps, "mAP": [float(np.mean(res)), float(np.mean(pres)), float(np.mean(aps))]}]

    def collect(self, imageid: str, det_box, det_mask, detfile: Dict[str, List[Dict]]) -> torch.Tensor:
        """the per-image body of both reference loops: paste this image's detections, append them to the
        per-class lists, return the merged class map (uint8 CUDA [H,W])"""
        image_h, image_w = self.sizes[imageid]
        entries, merged = paste_detections(det_box, det_mask, image_h, image_w, self.net_size)
        if entries and os.environ.get("DISYOLO_EVAL_GPU_IOU", "1") != "0":
            # round 6: the mask IoUs voc_eval needs (compute_overlaps_masks: every detection against the image's ground-truth
            # instances of its class) are taken HERE, on the GPU, from the pasted masks -- pixel counts as exact integers
            # ([nd, HW] x [HW, ng] in f32: every partial sum is an integer below 2^24), divided on the host in f32 exactly as
            # numpy does; the 0.5-MB-per-detection copies to the host and the numpy pass over them (145 ms per image) are gone
            for c in sorted({e["classid"] for e in entries}):
                dets = [e for e in entries if e["classid"] == c]
                gt = self._gt_stack(imageid, c, merged.device)
                if gt is None:
                    rows = [np.zeros(0, np.float32)] * len(dets)
                else:
                    d = torch.stack([e["mask"] for e in dets]).reshape(len(dets), -1).to(torch.float32)
                    inter = (d @ gt[0]).cpu().numpy()                                    # [nd, ng] f32, exact counts
                    union = d.sum(1).cpu().numpy()[:, None] + gt[1][None, :] - inter
                    rows = list((inter / union).astype(np.float32))
                for e, ov in zip(dets, rows):
                    detfile[str(c)].append({"imageid": imageid, "score": e["score"], "ov": ov})
            return merged
        for e in entries:
            detfile[str(e["classid"])].append({"imageid": imageid, "score": e["score"], "mask": e["mask"].cpu().numpy()})
        return merged

    def _gt_stack(self, imageid: str, classid: int, device):
        """the ground-truth instances of one class of one image, in ``recs_mask`` order (voc_eval's ``objs``): ([HW, ng] f32
        0 / 1 on the GPU, pixel counts [ng] f32 on the host); None without such instances.  Cached: a validation set is
        swept many times."""
        key = (imageid, classid)
        hit = self._gt_cache.get(key)
        if hit is None:
            objs = [o for o in self.recs_mask[imageid] if o["classid"] == classid]
            if not objs:
                hit = (None,)
            else:
                g = torch.from_numpy(np.stack([np.asarray(o["mask"]).astype(float) > 0.5 for o in objs]).reshape(len(objs), -1))
                g = g.to(device).to(torch.float32)
                hit = ((g.t().contiguous(), g.sum(1).cpu().numpy()),)
            self._gt_cache[key] = hit
        return hit[0]

    def do_python_eval(self, detdata: List[Dict]):
        """utils/validation_map.py:104-198: detdata = [{'boxes' [n,6], 'masks' [n,S,S] (CUDA tensor or numpy)
        or the scalar 0.0, 'imname'}] in ``index`` order -> [{'thresh', 'AP' [3], 'mAP' [recall, precision, mAP]}]"""
        assert len(detdata) == len(self.index)
        detfile = {str(c): [] for c in self.classid}
        for i, d in enumerate(detdata):
            assert d["imname"] == self.index[i]
            masks = d["masks"]
            if not torch.is_tensor(masks):
                if np.isscalar(masks) or np.ndim(masks) == 0 or np.sum(masks) == 0.0:
                    continue
                masks = torch.from_numpy(np.ascontiguousarray(masks, np.float32)).cuda()
            self.collect(d["imname"], d["boxes"], masks, detfile)
        return self._ap_table(detfile)


def evaluate(net, images: Dict[str, np.ndarray], eval_map: MAP, det_thresh: float = cfg.OBJ_THRESHOLD,
             weights_file: Optional[str] = None):
    """calculate_test_map.py:180-347.  ``images``: image id -> RGB uint8 array (or a path to decode);
    ``net``: a YOLONet built with batch size 1 like the reference's test graph (:354).  Returns
    (thresh_out, mask_acc, timing) = ([{'thresh', 'AP', 'mAP'}], [bg, crack, spall, rebar, mIoU] or None,
    {'prediction_s', 'crop_assemble_s', 'per_image_s'})."""
    if weights_file is not None:
        from .checkpoint import restore_net
        restore_net(net, weights_file)                       # saver.restore (:184-185)
    if net.B != 1:
        raise ValueError("evaluate() feeds one image at a time (cfg.BATCH_SIZE = 1, calculate_test_map.py:354)")
    S = net.S
    if (not net.training and getattr(net, "_infer_prog", None) is None and os.environ.get("DISYOLO_EVAL_REPLAY", "1") != "0"):
        # an inference net: record forward + detection filter + mask assembly once for this threshold (a hipGraph of one lane);
        # every image is then one replay (YOLONet.evaluation uses the recording when the threshold matches)
        net.build_infer_program(float(det_thresh), graph=True)
    detfile = {str(c): [] for c in eval_map.classid}
    seg = SegmentationAccuracy(net.device) if eval_map.merged is not None else None
    t_pred = t_crop = 0.0
    frame = torch.empty(1, S, S, 3, dtype=torch.float32, device=net.device)
    for index in eval_map.index:
        src = images[index]
        rgb = load_image_rgb(src) if isinstance(src, str) else np.asarray(src)
        image_h, image_w = rgb.shape[:2]
        assert [image_h, image_w] == list(eval_map.sizes[index])
        _, window = image_read(rgb, S, net.device, out=frame[0])
        torch.cuda.synchronize()
        t = time.time()
        det_box, det_mask = net.evaluation(frame, window[None], [np.float32(det_thresh)], masks_on_device=True)
        torch.cuda.synchronize()
        t_pred += time.time() - t
        t = time.time()
        if torch.is_tensor(det_mask[0]):
            merged = eval_map.collect(index, det_box[0], det_mask[0], detfile)
        else:                                                # np.sum(det_mask[0]) == 0.0 (:221-224)
            merged = torch.zeros(image_h, image_w, dtype=torch.uint8, device=net.device)
        if seg is not None:
            seg.add(eval_map.merged[index], merged)
        torch.cuda.synchronize()
        t_crop += time.time() - t
    thresh_out = eval_map._ap_table(detfile)
    mask_acc = seg.result() if seg is not None else None
    n = max(len(eval_map.index), 1)
    return thresh_out, mask_acc, {"prediction_s": t_pred, "crop_assemble_s": t_crop, "per_image_s": (t_pred + t_crop) / n}
