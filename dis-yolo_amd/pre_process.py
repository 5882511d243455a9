"""Dataset pre-processing: the counterpart of ``pre_process.py`` -- offline, once per
dataset, on the host: PASCAL-VOC XML boxes + one JPEG mask per class -> contours -> the ``ground_truth_cache``
records ``utils/train_data.py`` (here ``train_data.defect_train``) trains from.

The one non-trivial step of the reference is ``cv2.findContours(thresh, cv2.RETR_TREE, cv2.CHAIN_APPROX_NONE)``
(pre_process.py:78,82,86).  OpenCV is not installed in the build environment (neither interpreter), so the
border following is the library's own ``disyolo_find_contours`` (csrc/contours.hip, plain C++), checked against
the oracle's restatement of the published algorithm, hand-derived answers and topological invariants --
PARITY WITH cv2 ITSELF IS UNPINNED.  ``cv2.moments`` (Green's theorem) and the merge logic (:165-222) follow.
The verification drawing of the reference (``do_verification``, off in the shipped file) is not reproduced.
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import lib as L


def find_contours(binary: np.ndarray) -> Tuple[List[np.ndarray], np.ndarray]:
    """-> (contours: list of int32 [n,1,2] arrays of (x, y); hierarchy int32 [1,n,4]) like
    ``cv2.findContours(binary, cv2.RETR_TREE, cv2.CHAIN_APPROX_NONE)[1:]``"""
    img = np.ascontiguousarray(np.asarray(binary) != 0, dtype=np.uint8)
    if img.ndim != 2:
        raise ValueError("find_contours: a 2-D image is required")
    h, w = img.shape
    lib = L.load()
    nc, npts = C.c_int(0), C.c_int64(0)
    rc = lib.disyolo_find_contours(img.ctypes.data, h, w, None, 0, None, None, 0, C.byref(nc), C.byref(npts))
    if rc == 0 or nc.value == 0:
        return [], np.zeros((1, 0, 4), np.int32)
    pts = np.empty((npts.value, 2), np.int32)
    start = np.empty(nc.value + 1, np.int32)
    hier = np.empty((nc.value, 4), np.int32)
    rc = lib.disyolo_find_contours(img.ctypes.data, h, w, pts.ctypes.data, npts.value, start.ctypes.data, hier.ctypes.data,
                                   nc.value, C.byref(nc), C.byref(npts))
    if rc:
        raise L.DisyoloError("find_contours failed (%d): %s" % (rc, lib.disyolo_last_error().decode()))
    return [pts[start[k]:start[k + 1]].reshape(-1, 1, 2).copy() for k in range(nc.value)], hier[None]


def contour_centroid(points_xy) -> Tuple[int, int]:
    """(int(m10/m00), int(m01/m00)) of cv2.moments(contour): pre_process.py:177-180"""
    p = np.asarray(points_xy, np.float64).reshape(-1, 2)
    x0, y0 = p[:, 0], p[:, 1]
    x1, y1 = np.roll(x0, -1), np.roll(y0, -1)
    cross = x0 * y1 - x1 * y0
    a00, a10, a01 = cross.sum(), ((x0 + x1) * cross).sum(), ((y0 + y1) * cross).sum()
    if a00 == 0:
        raise ZeroDivisionError("float division by zero")       # the reference divides by m00 unguarded
    m00, m10, m01 = a00 / 2.0, a10 / 6.0, a01 / 6.0
    if m00 < 0:
        m00, m10, m01 = -m00, -m10, -m01
    return int(m10 / m00), int(m01 / m00)


def regions_from_masks(masks: Dict[str, Optional[np.ndarray]]) -> Tuple[Dict[str, Dict], int]:
    """pre_process.py:88-163.  masks: class name -> grey-level mask image (uint8) or None, thresholded at 127
    like ``cv2.threshold(img, 127, 255, 0)``; classes are visited in the reference's order crack, spall, rebar.
    -> (regions, number of contours nested two levels deep, which the reference counts as mask errors)"""
    regions: Dict[str, Dict] = {}
    count = errors = 0
    for classname in ("crack", "spall", "rebar"):
        img = masks.get(classname)
        if img is None:
            continue
        contours, hier = find_contours(np.asarray(img) > 127)
        pair: Dict[str, int] = {}
        for j, c in enumerate(contours):
            one = c[:, 0, :]
            all_x, all_y = one[:, 0].tolist(), one[:, 1].tolist()
            if hier[0, j, 3] == -1:
                regions[str(count)] = {"region_attributes": classname,
                                       "shape_attributes": [{"type": "out", "all_points_x": all_x, "all_points_y": all_y}]}
                pair[str(j)] = count
                count += 1
            else:
                parent = int(hier[0, j, 3])
                if hier[0, parent, 3] != -1:
                    errors += 1
                    continue
                regions[str(pair[str(parent)])]["shape_attributes"].append(
                    {"type": "in", "all_points_x": all_x, "all_points_y": all_y})
    return regions, errors


def merge_regions(regions: Dict[str, Dict], object_merge: Sequence[Sequence[float]], state: Optional[Dict] = None,
                  log=None) -> Dict[str, Dict]:
    """pre_process.py:165-222: the instances whose outer contour's centroid lies inside a 'merge' box of the XML
    annotation become one instance per box (closest box centre; class crack > rebar > spall).

    ``dis_index`` (the closest-box index) is a FUNCTION-level local of the reference's load_verify_contour: it survives
    from instance to instance AND from image to image (``state`` carries it; load_verify_contour passes one dict for
    the whole run), and the final containment test decides whether the stale box is used.  Two cases crash the
    reference (UnboundLocalError when no instance of any image so far had its centroid in a box; IndexError when the
    stale index does not exist in this image's box list): here the reference's own message for an unassigned
    instance is logged and the instance is skipped -- a data set the reference pre-processes is never aborted, and
    gives the same regions."""
    if not object_merge:
        return {}
    if state is None:
        state = {}
    groups = {jj: [] for jj in range(len(object_merge))}
    names = {jj: [] for jj in range(len(object_merge))}
    for k in range(len(regions)):
        reg = regions[str(k)]
        poly = reg["shape_attributes"][0]
        cX, cY = contour_centroid(np.column_stack([poly["all_points_x"], poly["all_points_y"]]))
        old = 4000
        for ii, (x1, y1, x2, y2) in enumerate(object_merge):
            if cX <= x1 or cX >= x2 or cY <= y1 or cY >= y2:
                continue
            d = (((x1 + x2) / 2 - cX) ** 2 + ((y1 + y2) / 2 - cY) ** 2) ** 0.5
            if d < old:
                state["dis_index"], old = ii, d
        dis_index = state.get("dis_index")
        assigned = False
        if dis_index is not None and dis_index < len(object_merge):
            x1, y1, x2, y2 = object_merge[dis_index]
            if x1 <= cX <= x2 and y1 <= cY <= y2:
                groups[dis_index].extend(reg["shape_attributes"])
                names[dis_index].append(reg["region_attributes"])
                assigned = True
        if not assigned and log is not None:
            log("No merged box belongs to the defect")
    new_regions, count = {}, 0
    for jj in range(len(object_merge)):
        if not groups[jj]:
            continue
        nl = names[jj]
        cls = "crack" if "crack" in nl else ("spall" if ("spall" in nl and "rebar" not in nl) else "rebar")
        new_regions[str(count)] = {"region_attributes": cls, "shape_attributes": groups[jj]}
        count += 1
    return new_regions


def parse_voc_boxes(xml_path: str) -> List[Dict]:
    """pre_process.py:46-60: objects of a PASCAL-VOC file, boxes shifted to 0-based pixel coordinates"""
    objects = []
    for obj in ET.parse(xml_path).findall("object"):
        bb = obj.find("bndbox")
        objects.append({"class": obj.find("name").text.lower().strip(),
                        "bbox": [float(bb.find(k).text) - 1 for k in ("xmin", "ymin", "xmax", "ymax")]})
    return objects


def _read_grey(path: str) -> np.ndarray:
    from PIL import Image        # (the reference: cv2.imread(..., cv2.IMREAD_GRAYSCALE))
    return np.asarray(Image.open(path).convert("L"))


def load_verify_contour(data_path: str, phase: str = "train", log=print) -> List[Dict]:
    """pre_process.py:16-318 without the verification drawing: builds (or loads) ``cache/ground_truth_cache.pkl``
    = [{'filename', 'regions', 'size': [h, w]}] and ``cache/<phase>.txt``"""
    from PIL import Image
    root = os.path.join(data_path, phase)
    cache_dir = os.path.join(root, "cache")
    cache = os.path.join(cache_dir, "ground_truth_cache.pkl")
    if os.path.isfile(cache):
        log("Loading gt_labels from: " + cache)
        with open(cache, "rb") as f:
            return pickle.load(f)
    os.makedirs(cache_dir, exist_ok=True)
    annotations, error_mask = [], 0
    merge_state: Dict = {}        # the reference's function-level dis_index (pre_process.py:192): survives from image to image
    with open(os.path.join(cache_dir, phase + ".txt"), "w") as ids:
        for file in os.listdir(os.path.join(root, "images")):
            name = os.path.splitext(file)[0]
            log(name)
            ids.write(name + "\n")
            xml = os.path.join(root, "annotations", name + ".xml")
            has_xml = os.path.exists(xml)
            merge = [o["bbox"] for o in parse_voc_boxes(xml) if o["class"] == "merge"] if has_xml else []
            masks = {}
            for cls in ("crack", "spall", "rebar"):
                p = os.path.join(root, "masks", name + cls + ".jpg")
                masks[cls] = _read_grey(p) if os.path.exists(p) else None
            regions, errors = regions_from_masks(masks)
            error_mask += errors
            if has_xml:
                regions = merge_regions(regions, merge, merge_state, lambda m: log(m + " in " + file))
            with Image.open(os.path.join(root, "images", name + ".jpg")) as im:
                width, height = im.size
            annotations.append({"filename": file, "regions": regions, "size": [height, width]})
    log("Number of error mask is " + str(error_mask))
    log("Saving gt_labels to: " + cache)
    with open(cache, "wb") as f:
        pickle.dump(annotations, f)
    return annotations
