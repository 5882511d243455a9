"""Generate the golden fixtures under tests/golden/ by importing the reference's own
importable modules (yolo/config.py, utils/voc_eval_mask.py) from /root/reference.
Run in the build container only (the reference never travels to the GPU box):

    python tools/make_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, REF)


def config_fixture():
    import yolo.config as c
    d = {}
    for k in sorted(c.__dict__):
        if k[0].isupper() and k not in ("MODEL_PATH", "DATASET", "OUTPUT_DIR", "WEIGHTS_FILE"):
            v = getattr(c, k)
            d[k] = v.tolist() if isinstance(v, np.ndarray) else v
    with open(os.path.join(OUT, "config.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


def voc_fixture():
    from utils.voc_eval_mask import voc_eval, voc_ap, compute_overlaps_masks
    rng = np.random.RandomState(7)
    H = W = 24
    cases = []
    for case in range(6):
        nimg = rng.randint(1, 4)
        names = ["img%d" % i for i in range(nimg)]
        recs, gts = {}, {}
        for n in names:
            objs = []
            for _ in range(rng.randint(0, 4)):
                m = np.zeros((H, W), np.uint8)
                y, x = rng.randint(0, H - 8), rng.randint(0, W - 8)
                h, w = rng.randint(3, 8), rng.randint(3, 8)
                m[y:y + h, x:x + w] = 1
                objs.append({"classid": int(rng.randint(0, 2)), "mask": m, "difficult": int(rng.rand() < 0.15)})
            recs[n] = objs
        dets = []
        for n in names:
            for o in recs[n]:
                if rng.rand() < 0.75:
                    m = np.roll(o["mask"], rng.randint(-2, 3), axis=rng.randint(0, 2))
                    dets.append({"imageid": n, "score": float(np.round(rng.rand(), 3)), "mask": m, "classid": o["classid"]})
            for _ in range(rng.randint(0, 3)):
                m = np.zeros((H, W), np.uint8)
                y, x = rng.randint(0, H - 6), rng.randint(0, W - 6)
                m[y:y + 5, x:x + 5] = 1
                dets.append({"imageid": n, "score": float(np.round(rng.rand(), 3)), "mask": m,
                             "classid": int(rng.randint(0, 2))})
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write("\n".join(names) + "\n")
            setfile = f.name
        results = {}
        for cid in (0, 1):
            for use07 in (False, True):
                detfile = [d for d in dets if d["classid"] == cid]
                r = voc_eval(detfile, recs, setfile, cid, ovthresh=0.5, use_07_metric=use07)
                results["%d_%d" % (cid, int(use07))] = [float(np.asarray(v)) for v in r]
        os.unlink(setfile)
        cases.append({"names": names,
                      "recs": {n: [{"classid": o["classid"], "difficult": o["difficult"], "mask": o["mask"].tolist()}
                                   for o in recs[n]] for n in names},
                      "dets": [{"imageid": d["imageid"], "score": d["score"], "classid": d["classid"],
                                "mask": d["mask"].tolist()} for d in dets],
                      "results": results})
    # the survey's hand-checked known answer: 3 detections / 2 GT -> (1.0, 0.6667, 0.8333)
    ap_cases = []
    for _ in range(5):
        n = rng.randint(1, 8)
        rec = np.sort(rng.rand(n))
        prec = rng.rand(n)
        ap_cases.append({"rec": rec.tolist(), "prec": prec.tolist(), "ap": float(voc_ap(rec, prec, False)),
                         "ap07": float(voc_ap(rec, prec, True))})
    m1 = (rng.rand(10, 10, 3) > 0.5).astype(np.float32)
    m2 = (rng.rand(10, 10, 2) > 0.5).astype(np.float32)
    ov = compute_overlaps_masks(m1, m2)
    with open(os.path.join(OUT, "voc_eval.json"), "w") as f:
        json.dump({"cases": cases, "ap_cases": ap_cases,
                   "overlaps": {"m1": m1.tolist(), "m2": m2.tolist(), "iou": ov.tolist()}}, f)


def boxes_fixture():
    """calculate_test_map.py cannot be imported (tensorflow, cv2 at module level), but its box
    un-letterboxing method `correct_yolo_boxes` (:121-138) is plain numpy arithmetic: the
    function is taken out of the reference's own file with `ast` and executed here, in this
    container only, to produce input/output vectors for the four sample image sizes."""
    import ast
    src = open(os.path.join(REF, "calculate_test_map.py")).read()
    fn = None
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name == "correct_yolo_boxes":
            fn = node
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, "calculate_test_map.py", "exec"), ns)
    ref = ns["correct_yolo_boxes"]
    from PIL import Image
    import glob
    sizes = sorted({Image.open(f).size[::-1] for f in glob.glob(os.path.join(REF, "data/train_sample/images/*.jpg"))})
    rng = np.random.RandomState(3)
    cases = []
    for (h, w) in sizes:
        for net in (576, 832):
            boxes = rng.rand(24, 4).astype(np.float32)
            boxes[:4] = [[0, 0, 1, 1], [0.5, 0.5, 0.5, 0.5], [0.25, 0.1, 0.2, 0.9], [0.999, 0.001, 1.0, 0.0]]
            for (x1, y1, x2, y2) in boxes:
                r = ref(None, np.float32(x1), np.float32(y1), np.float32(x2), np.float32(y2), h, w, net, net)
                cases.append({"box": [float(x1), float(y1), float(x2), float(y2)], "image_hw": [int(h), int(w)],
                              "net": net, "out": [int(v) for v in r]})
    with open(os.path.join(OUT, "correct_yolo_boxes.json"), "w") as f:
        json.dump({"source": "calculate_test_map.py:121-138 executed by tools/make_golden.py", "cases": cases}, f)


def miou_fixture():
    """the semantic-segmentation accuracy block of `evaluate` (calculate_test_map.py:303-346) is
    plain numpy on two dicts of class maps: its statements are taken out of the reference's file
    with `ast` and executed here on random class maps."""
    import ast
    src = open(os.path.join(REF, "calculate_test_map.py")).read()
    fn = [n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "evaluate"][0]
    body = [s for s in fn.body if 303 <= s.lineno <= 346]
    code = compile(ast.Module(body=body, type_ignores=[]), "calculate_test_map.py", "exec")
    rng = np.random.RandomState(11)
    cases = []
    for case in range(5):
        names = ["im%d" % i for i in range(rng.randint(1, 4))]
        true, pred = {}, {}
        for n in names:
            h, w = rng.randint(5, 12), rng.randint(5, 12)
            true[n] = rng.randint(0, 4, size=(h, w)).astype(np.uint8)
            pred[n] = np.where(rng.rand(h, w) < 0.6, true[n], rng.randint(0, 4, size=(h, w))).astype(np.uint8)
        ns = {"np": np, "val_index": names, "val_mergemask": true, "det_masks": pred}
        exec(code, ns)
        cases.append({"names": names, "true": {n: true[n].tolist() for n in names},
                      "pred": {n: pred[n].tolist() for n in names}, "mask_acc": [float(v) for v in ns["mask_acc"]]})
    with open(os.path.join(OUT, "miou.json"), "w") as f:
        json.dump({"source": "calculate_test_map.py:303-346 executed by tools/make_golden.py", "cases": cases}, f)


def targets_fixture():
    """utils/train_data.py cannot be imported (cv2, pyblur, skimage), but the loop that encodes the
    ground-truth boxes into the three YOLO target grids (:134-178, "prepare training input for
    yolos") is plain numpy: the `for index in bbox_index:` statement is taken out of the file with
    `ast` and executed here with the surrounding locals set to the no-augmentation case
    (sx = sy = 1, dx = dy = 0)."""
    import ast
    import types
    import yolo.config as c
    src = open(os.path.join(REF, "utils", "train_data.py")).read()
    loop = None
    for node in ast.walk(ast.parse(src)):
        if (isinstance(node, ast.For) and isinstance(node.target, ast.Name) and node.target.id == "index"
                and isinstance(node.iter, ast.Name) and node.iter.id == "bbox_index" and 130 <= node.lineno <= 140):
            loop = node
    code = compile(ast.Module(body=[loop], type_ignores=[]), "train_data.py", "exec")
    rng = np.random.RandomState(21)
    cases = []
    for net in (576, 832):
        for case in range(4):
            n = rng.randint(1, 9)
            bbox = np.zeros((1, 1, 1, 20, 5), np.float32)
            for j in range(n):
                w, h = rng.uniform(4, 0.7 * net), rng.uniform(4, 0.7 * net)
                if case == 3 and j > 0:          # collisions: same cell and anchor as box 0
                    x1, y1 = bbox[0, 0, 0, 0, 0] + rng.uniform(-2, 2), bbox[0, 0, 0, 0, 1] + rng.uniform(-2, 2)
                    w, h = bbox[0, 0, 0, 0, 2] - bbox[0, 0, 0, 0, 0], bbox[0, 0, 0, 0, 3] - bbox[0, 0, 0, 0, 1]
                else:
                    x1, y1 = rng.uniform(0, net - w), rng.uniform(0, net - h)
                bbox[0, 0, 0, j] = [x1, y1, x1 + w, y1 + h, rng.randint(0, 3)]
            inp = bbox[0, 0, 0, :n].copy()
            g = net // 32
            yolos = [np.zeros((4 * g, 4 * g, 3, 8), np.float32), np.zeros((2 * g, 2 * g, 3, 8), np.float32),
                     np.zeros((g, g, 3, 8), np.float32)]
            ns = {"np": np, "self": types.SimpleNamespace(anchors=c.ANCHORS, num_anchor=3), "bbox_index": list(range(n)),
                  "bbox": bbox, "sx": 1.0, "sy": 1.0, "dx": 0, "dy": 0, "net_w": net, "net_h": net, "yolos": yolos,
                  "print": lambda *a: None}
            exec(code, ns)
            nz = [{"grid": gi, "idx": [int(v) for v in idx], "row": yolos[gi][tuple(idx)].tolist()}
                  for gi in range(3) for idx in np.argwhere(yolos[gi][..., 4] == 1)]
            cases.append({"net": net, "boxes_x1y1x2y2c": inp.tolist(), "true_box_xcycwh": bbox[0, 0, 0, :n, :4].tolist(),
                          "objects": nz})
    with open(os.path.join(OUT, "assign_targets.json"), "w") as f:
        json.dump({"source": "utils/train_data.py:134-178 executed by tools/make_golden.py", "cases": cases}, f)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    config_fixture()
    voc_fixture()
    boxes_fixture()
    miou_fixture()
    targets_fixture()
    print("golden fixtures written to", OUT)
