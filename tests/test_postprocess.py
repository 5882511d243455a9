"""evaluate()'s host-side post-processing: the box un-letterboxing is pinned by golden
vectors produced by the reference's own function; the GPU mask paste is checked against the
oracle's restatement of the cv2-based loop."""
import json
import os

import numpy as np
import pytest
import torch

import disyolo_oracle as O
from disyolo_amd import postprocess as P

GOLD = os.path.join(os.path.dirname(__file__), "golden", "correct_yolo_boxes.json")
GOLD_MIOU = os.path.join(os.path.dirname(__file__), "golden", "miou.json")


def test_segmentation_miou_matches_reference_golden():
    for c in json.load(open(GOLD_MIOU))["cases"]:
        got = O.segmentation_miou([np.array(c["true"][n]) for n in c["names"]], [np.array(c["pred"][n]) for n in c["names"]])
        np.testing.assert_allclose(got, c["mask_acc"], rtol=1e-12)


@pytest.mark.gpu
def test_gpu_confusion_counts_match_reference_golden(dev):
    for c in json.load(open(GOLD_MIOU))["cases"]:
        acc = P.SegmentationAccuracy(dev)
        for n in c["names"]:
            acc.add(np.array(c["true"][n], np.uint8), torch.tensor(c["pred"][n], dtype=torch.uint8, device=dev))
        np.testing.assert_allclose(acc.result(), c["mask_acc"], rtol=1e-12)
    # a full-size pair: counts are exact integers whatever the block scheduling
    g = torch.Generator(device=dev).manual_seed(0)
    t = torch.randint(0, 4, (754, 1008), device=dev, generator=g, dtype=torch.uint8)
    p = torch.randint(0, 4, (754, 1008), device=dev, generator=g, dtype=torch.uint8)
    acc = P.SegmentationAccuracy(dev)
    acc.add(t, p)
    want = torch.bincount(t.flatten().long() * 4 + p.flatten().long(), minlength=16)
    assert torch.equal(acc.conf, want)


def test_correct_yolo_boxes_matches_reference_golden():
    cases = json.load(open(GOLD))["cases"]
    assert len(cases) >= 100
    for c in cases:
        h, w = c["image_hw"]
        x1, y1, x2, y2 = c["box"]
        assert list(O.correct_yolo_boxes(x1, y1, x2, y2, h, w, c["net"], c["net"])) == c["out"]
        got = P.correct_yolo_boxes(np.array([[y1, x1, y2, x2]], np.float32), h, w, c["net"], c["net"])[0]
        assert got.tolist() == c["out"]


def test_letterbox_window_and_resize_known_answers():
    # 620x348 image in a 576 box: resized to 576x323, 126 rows of padding above
    np.testing.assert_allclose(O.letterbox_window(348, 620, 576), [126 / 576, 0, (323 + 126) / 576, 1.0], rtol=1e-6)
    np.testing.assert_array_equal(P.letterbox_window(600, 800, 576), O.letterbox_window(600, 800, 576))
    x = np.arange(12, dtype=np.float32).reshape(3, 4)
    np.testing.assert_array_equal(O.resize_linear(x, 4, 3), x)                  # identity
    up = O.resize_linear(x, 8, 6)                                               # 2x: centres at -0.25, 0.25, ...
    np.testing.assert_allclose(up[0], [0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.0])
    np.testing.assert_allclose(up[:, 0], [0, 1, 3, 5, 7, 8])
    np.testing.assert_allclose(O.resize_linear(x, 2, 1), [[4.5, 6.5]])          # 2x2 box means at the centre row


@pytest.mark.gpu
def test_mask_paste_matches_oracle(dev):
    rng = np.random.RandomState(5)
    size, net = 96, 192
    for (h, w) in ((348, 620), (450, 386), (192, 192)):
        n = 9
        masks = rng.rand(n, size, size).astype(np.float32)
        box = np.zeros((n, 6), np.float32)
        win = O.letterbox_window(h, w, net)
        for k in range(n):
            y1, x1 = rng.uniform(win[0], win[2] - 0.05), rng.uniform(win[1], win[3] - 0.05)
            box[k, :4] = [y1, x1, min(y1 + rng.uniform(0.02, 0.6), 1.0), min(x1 + rng.uniform(0.02, 0.6), 1.0)]
            box[k, 4], box[k, 5] = rng.randint(0, 3), rng.rand()
        box[3, :4] = [0.5, 0.5, 0.5, 0.7]          # empty box: skipped
        box[4, :4] = [0.0, 0.0, 1.0, 1.0]          # whole letter box, clamped to the image
        want_entries, want_merged = O.paste_detections(box, masks, h, w, net)
        entries, merged = P.paste_detections(box, torch.from_numpy(masks).to(dev), h, w, net)
        torch.cuda.synchronize()
        assert [e["index"] for e in entries] == [e["index"] for e in want_entries]
        for e, we in zip(entries, want_entries):
            assert e["classid"] == we["classid"]
            np.testing.assert_array_equal(e["mask"].cpu().numpy(), we["mask"])
        np.testing.assert_array_equal(merged.cpu().numpy(), want_merged)
    # an image without detections: the scalar 0.0 of `evaluation`
    entries, merged = P.paste_detections(np.zeros((0, 6), np.float32), 0.0, 50, 60, net)
    assert entries == [] and int(merged.sum()) == 0


@pytest.mark.gpu
def test_evaluation_to_pasted_masks_end_to_end(dev):
    """evaluation(masks_on_device=True) -> paste_detections == the oracle's loop on the same
    evaluation output (a 64x64 net, boosted heads so that there are detections)."""
    from disyolo_amd.net import YOLONet
    net = YOLONet(training=False, device=dev, image_size=64, batch_size=2, stage=1, seed=3)
    with torch.no_grad():
        for i in (59, 67, 75, 82):
            net.params["yolo/convolutional%d/weights" % i].mul_(40.0)
    net.refresh_weights()
    b = O.synthetic_batch(2, 64, seed=9)
    h, w = 90, 120
    win = np.tile(O.letterbox_window(h, w, 64)[None], (2, 1))
    det_box, det_mask = net.evaluation(b["images"], win, 0.05, masks_on_device=True)
    torch.cuda.synchronize()
    assert sum(len(d) for d in det_box) > 0
    for i in range(2):
        entries, merged = P.paste_detections(det_box[i], det_mask[i], h, w, 64)
        host_mask = det_mask[i].cpu().numpy() if torch.is_tensor(det_mask[i]) else det_mask[i]
        want_entries, want_merged = O.paste_detections(det_box[i], host_mask, h, w, 64)
        assert [e["index"] for e in entries] == [e["index"] for e in want_entries]
        for e, we in zip(entries, want_entries):
            np.testing.assert_array_equal(e["mask"].cpu().numpy(), we["mask"])
        np.testing.assert_array_equal(merged.cpu().numpy(), want_merged)
