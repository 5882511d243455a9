"""Dataset pre-processing (pre_process.py:16-318): contour extraction + region building.
cv2 is not installable here, so nothing in this file compares with OpenCV itself (parity unpinned): the native
border following is held to (i) hand-derived answers that follow from the documented conventions of
cv2.findContours (outer borders from their top-left pixel down the left side, holes the other way round, every
border pixel visit listed, siblings newest first, [next, previous, first_child, parent]), (ii) the oracle's
independent restatement on random images, (iii) topological invariants computed with scipy.ndimage."""
import json
import os
import pickle

import numpy as np
import pytest
from scipy import ndimage

import disyolo_oracle as O
from disyolo_amd import pre_process as P


def _img(rows):
    return np.array([[1 if c == "#" else 0 for c in r] for r in rows], np.uint8)


def _lists(contours):
    return [c.reshape(-1, 2).tolist() for c in contours]


def test_known_answers():
    c, h = P.find_contours(_img([".....", ".###.", ".###.", ".###.", "....."]))
    assert _lists(c) == [[[1, 1], [1, 2], [1, 3], [2, 3], [3, 3], [3, 2], [3, 1], [2, 1]]]
    assert h.tolist() == [[[-1, -1, -1, -1]]]
    c, h = P.find_contours(_img(["...", ".#.", "..."]))
    assert _lists(c) == [[[1, 1]]] and h.shape == (1, 1, 4)
    # a one-pixel-wide horizontal line: every pixel but the ends is visited twice (there and back)
    c, _ = P.find_contours(_img([".....", ".###.", "....."]))
    assert _lists(c) == [[[1, 1], [2, 1], [3, 1], [2, 1]]]
    # ring: the hole's border is made of the ring's pixels around it, listed the other way round
    c, h = P.find_contours(_img([".......", ".#####.", ".#####.", ".##.##.", ".#####.", ".#####.", "......."]))
    assert len(c) == 2 and _lists(c)[1] == [[2, 3], [3, 2], [4, 3], [3, 4]]
    assert h[0].tolist() == [[-1, -1, 1, -1], [-1, -1, -1, 0]]
    # two objects: the one found later in the scan comes first
    c, h = P.find_contours(_img([".........", ".##......", ".##..###.", ".....###.", "........."]))
    assert _lists(c)[0][0] == [5, 2] and _lists(c)[1][0] == [1, 1]
    assert h[0].tolist() == [[1, -1, -1, -1], [-1, 0, -1, -1]]
    # empty image
    c, h = P.find_contours(np.zeros((4, 6), np.uint8))
    assert c == [] and h.shape == (1, 0, 4)
    # diagonal neighbours belong to one object (8-connectivity)
    c, _ = P.find_contours(_img(["....", ".#..", "..#.", "...."]))
    assert len(c) == 1


def _random_image(rng, h, w):
    a = (ndimage.gaussian_filter(rng.standard_normal((h, w)), 2.0) > 0.02).astype(np.uint8)
    if rng.rand() < 0.5:                      # an object inside a hole inside an object
        a[3:h - 3, 3:w - 3] = 1
        a[6:h - 6, 6:w - 6] = 0
        a[9:h - 9, 9:w - 9] |= (rng.rand(max(h - 18, 0), max(w - 18, 0)) > 0.5).astype(np.uint8)
    return a


@pytest.mark.parametrize("seed", range(12))
def test_native_tracer_equals_the_oracle_and_the_topology(seed):
    rng = np.random.RandomState(seed)
    h, w = rng.randint(20, 48), rng.randint(20, 64)
    a = _random_image(rng, h, w)
    contours, hier = P.find_contours(a)
    want_c, want_h = O.find_contours_tree(a)
    assert len(contours) == len(want_c)
    for got, want in zip(contours, want_c):
        assert np.array_equal(got.reshape(-1, 2), want)
    assert np.array_equal(hier[0], want_h)
    # every listed point is a foreground pixel with a background 4-neighbour or on the image edge
    pad = np.pad(a, 1)
    for c in contours:
        x, y = c[:, 0, 0], c[:, 0, 1]
        assert a[y, x].all()
        nb = np.stack([pad[y, x + 1], pad[y + 2, x + 1], pad[y + 1, x], pad[y + 1, x + 2]])
        diag = np.stack([pad[y, x], pad[y, x + 2], pad[y + 2, x], pad[y + 2, x + 2]])
        assert ((nb == 0).any(0) | (diag == 0).any(0)).all()
    # depth parity: outer borders <-> 8-connected foreground components, holes <-> enclosed 4-connected background
    depth = np.zeros(len(contours), int)
    for k in range(len(contours)):
        p, d = hier[0, k, 3], 0
        while p != -1:
            p, d = hier[0, p, 3], d + 1
        depth[k] = d
    n_fg = ndimage.label(a, structure=np.ones((3, 3)))[1]
    lab, n_bg = ndimage.label(pad == 0)
    n_holes = n_bg - 1                                   # all but the outside
    assert (depth % 2 == 0).sum() == n_fg and (depth % 2 == 1).sum() == n_holes
    # hierarchy links are mutually consistent
    for k in range(len(contours)):
        nxt, prv, child, par = hier[0, k]
        if nxt != -1:
            assert hier[0, nxt, 1] == k and hier[0, nxt, 3] == par
        if child != -1:
            assert hier[0, child, 3] == k and hier[0, child, 1] == -1


def test_centroid_and_regions_follow_the_oracle():
    assert P.contour_centroid([[1, 1], [1, 3], [3, 3], [3, 1]]) == O.contour_centroid(np.array([[1, 1], [1, 3], [3, 3], [3, 1]])) == (2, 2)
    with pytest.raises(ZeroDivisionError):
        P.contour_centroid([[4, 4]])
    rng = np.random.RandomState(5)
    masks = {}
    for cls in ("crack", "spall", "rebar"):
        a = np.zeros((60, 80), np.uint8)
        for _ in range(3):
            y, x = rng.randint(5, 45), rng.randint(5, 60)
            a[y:y + rng.randint(4, 12), x:x + rng.randint(4, 16)] = 255
        a[20:24, 30:33] = 0                                    # may punch a hole
        masks[cls] = a
    masks["spall"][10:40, 10:50] = 255                         # object with an object in its hole: nested two deep
    masks["spall"][15:35, 15:45] = 0
    masks["spall"][20:30, 20:40] = 255
    masks["spall"][23:27, 25:35] = 0
    got, errs = P.regions_from_masks(masks)
    per_class = []
    for cls in ("crack", "spall", "rebar"):
        c, h = O.find_contours_tree(masks[cls] > 127)
        per_class.append((cls, c, h))
    want, want_errs = O.regions_from_contours(per_class)
    assert got == want and errs == want_errs and errs >= 1
    boxes = [[-1.0, -1.0, 80.0, 60.0], [40.0, 20.0, 79.0, 59.0], [2.0, 2.0, 30.0, 30.0]]
    merged = P.merge_regions(got, boxes)
    assert merged == O.merge_regions(want, boxes) and 1 <= len(merged) <= 3
    assert sum(len(r["shape_attributes"]) for r in merged.values()) == sum(len(r["shape_attributes"]) for r in got.values())
    assert P.merge_regions(got, []) == {}
    # first centroid in no box: the reference crashes (UnboundLocalError); here its message is logged and the instance skipped
    msgs = []
    assert P.merge_regions(got, [[70.0, 50.0, 75.0, 55.0]], {}, msgs.append) == {} and len(msgs) == len(got)
    # the closest-box index is function-level state of the reference: it survives from image to image, and the final
    # containment test decides whether the stale box takes the instance
    state = {}
    first = P.merge_regions(got, boxes, state)
    assert first == merged and state["dis_index"] is not None
    stale = state["dis_index"]
    far = [[500.0, 500.0, 510.0, 510.0]] * (stale + 1)          # no centroid in any of these boxes
    assert P.merge_regions(got, far, state) == {} and state["dis_index"] == stale
    assert P.merge_regions(got, far[:1], {"dis_index": 7}) == {}        # stale index beyond this image's box list


def test_load_verify_contour_builds_the_cache(tmp_path):
    from PIL import Image
    root = tmp_path / "data" / "train"
    for d in ("images", "masks", "annotations"):
        (root / d).mkdir(parents=True)
    rgb = (np.random.RandomState(0).rand(48, 64, 3) * 255).astype(np.uint8)
    Image.fromarray(rgb).save(root / "images" / "a.jpg")
    m = np.zeros((48, 64), np.uint8)
    m[10:30, 12:40] = 255
    m[16:22, 20:30] = 0
    Image.fromarray(m).save(root / "masks" / "aspall.jpg", quality=100)
    (root / "annotations" / "a.xml").write_text(
        "<annotation><object><name>Merge</name><bndbox><xmin>5</xmin><ymin>5</ymin><xmax>60</xmax><ymax>44</ymax></bndbox></object>"
        "<object><name>spall</name><bndbox><xmin>13</xmin><ymin>11</ymin><xmax>40</xmax><ymax>30</ymax></bndbox></object></annotation>")
    ann = P.load_verify_contour(str(tmp_path / "data"), "train", log=lambda s: None)
    assert len(ann) == 1 and ann[0]["filename"] == "a.jpg" and ann[0]["size"] == [48, 64]
    reg = ann[0]["regions"]
    assert list(reg) == ["0"] and reg["0"]["region_attributes"] == "spall"
    shapes = reg["0"]["shape_attributes"]
    assert [s["type"] for s in shapes] == ["out", "in"]
    assert min(shapes[0]["all_points_x"]) == 12 and max(shapes[0]["all_points_x"]) == 39
    assert min(shapes[0]["all_points_y"]) == 10 and max(shapes[0]["all_points_y"]) == 29
    with open(root / "cache" / "ground_truth_cache.pkl", "rb") as f:
        assert pickle.load(f) == ann
    assert (root / "cache" / "train.txt").read_text() == "a\n"
    assert P.load_verify_contour(str(tmp_path / "data"), "train", log=lambda s: None) == ann      # second call: from the cache


def test_reference_sample_data_golden_and_synthetic_replay():
    """A REGRESSION fixture, not a pin: tests/golden/pre_process_sample.json is what THIS package's load_verify_contour
    produced on the reference's data/train_sample -- the reference's pre_process.py needs cv2 and never ran, so nothing
    here is reference output (cv2.findContours parity stays unpinned).  (Four images, six class masks, 00044.xml with
    four 'merge' boxes; tools/make_golden_pre_process.py, build container only -- the JPEGs do not travel.)  (1) what the file records must be self-consistent: the contour tracer's outer
    borders / holes equal scipy.ndimage's independent component counts of the same masks, and the four merge boxes
    of 00044.xml became four merged instances (three rebar pieces in the first box).  (2) replay: synthetic masks
    with one small blob at every recorded instance centroid go through regions_from_masks + merge_regions with the
    recorded boxes and must reproduce the merged instances (class, number of polygons, box)."""
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pre_process_sample.json")))
    assert [im["filename"] for im in g["images"]] == ["00044.jpg", "00054.jpg", "001005.jpg", "01015.jpg"]
    for im in g["images"]:
        before = im["instances_before_merge"]
        for cls, ind in im["independent"].items():
            mine = [b for b in before if b["class"] == cls]
            assert len(mine) == ind["components_8"], (im["filename"], cls)          # outer borders <-> 8-connected components
            assert sum(b["holes"] for b in mine) == ind["holes"]
        assert im["mask_errors"] == 0
        H, W = im["size_hw"]
        for inst in im["instances"]:
            x1, y1, x2, y2 = inst["bbox"]
            assert 0 <= x1 <= x2 < W and 0 <= y1 <= y2 < H
    im44 = g["images"][0]
    assert len(im44["merge_boxes"]) == 4 and len(im44["instances_before_merge"]) == 6
    assert [(i["class"], i["polygons"]) for i in im44["instances"]] == [("rebar", 3), ("spall", 1), ("rebar", 1), ("rebar", 1)]
    assert [len(im["instances"]) for im in g["images"][1:]] == [2, 2, 1]           # no XML: instances pass through
    # ---- replay on synthetic masks of the same topology
    state = {}
    for im in g["images"]:
        H, W = im["size_hw"]
        masks = {c: None for c in ("crack", "spall", "rebar")}
        for b in im["instances_before_merge"]:
            if masks[b["class"]] is None:
                masks[b["class"]] = np.zeros((H, W), np.uint8)
            cx, cy = b["centroid"]
            masks[b["class"]][max(cy - 3, 0):cy + 4, max(cx - 3, 0):cx + 4] = 255
        regions, errors = P.regions_from_masks(masks)
        assert errors == 0 and len(regions) == len(im["instances_before_merge"])
        if im["merge_boxes"] is not None:
            regions = P.merge_regions(regions, im["merge_boxes"], state)
        assert [(r["region_attributes"], len(r["shape_attributes"])) for r in (regions[str(k)] for k in range(len(regions)))] == \
            [(i["class"], i["polygons"]) for i in im["instances"]], im["filename"]
