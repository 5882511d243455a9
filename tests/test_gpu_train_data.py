"""GPU kernels of the training-data pipeline (csrc/augment.hip) against the oracle, bit for bit, and the
``defect_train.get()`` mirror (utils/train_data.py:44-276) replayed by the oracle with the same decisions."""
import json
import os

import numpy as np
import pytest
import torch

import disyolo_oracle as O
from disyolo_amd import config as cfg
from disyolo_amd import lib as L
from disyolo_amd import train_data as TD

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "polygon.json")


def test_polygon_kernel_matches_scikit_image_golden(dev):
    g = json.load(open(GOLD))
    for c in g["cases"]:
        H, W = max(c["y"]) + 3, max(c["x"]) + 3
        m = TD.rasterize_instance([{"type": "out", "all_points_x": c["x"], "all_points_y": c["y"]}], H, W, dev)
        torch.cuda.synchronize()
        want = np.zeros((H, W), bool)
        want[c["rr"], c["cc"]] = True                   # skimage.draw.polygon
        want[c["y"], c["x"]] = True                     # + the vertex pixels (utils/train_data.py:333)
        np.testing.assert_array_equal(m.cpu().numpy().astype(bool), want, err_msg=c["name"])


def test_polygon_kernel_holes_order_and_oracle(dev):
    rng = np.random.RandomState(3)
    for trial in range(6):
        H, W = 60 + trial * 7, 80 - trial * 5
        polys = []
        for k in range(rng.randint(1, 5)):
            n = rng.randint(3, 8)
            polys.append({"type": "out" if (k == 0 or rng.rand() < 0.5) else "in",
                          "all_points_x": rng.randint(0, W, n).tolist(), "all_points_y": rng.randint(0, H, n).tolist()})
        m = TD.rasterize_instance(polys, H, W, dev)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(m.cpu().numpy().astype(bool), O.instance_mask(polys, H, W))


PLACE = [  # image H, W, size, new_w, new_h, dx, dy
    (40, 50, 64, 64, 51, 0, 6), (40, 50, 64, 48, 38, 9, 20), (33, 21, 64, 90, 141, -13, -40), (50, 40, 32, 20, 25, 3, 2),
    (17, 64, 64, 64, 17, 0, 23), (30, 30, 64, 95, 95, -31, -16)]


@pytest.mark.parametrize("H,W,S,nw,nh,dx,dy", PLACE)
@pytest.mark.parametrize("flip", [1, 2, 3])
def test_place_image_and_mask_match_oracle(dev, H, W, S, nw, nh, dx, dy, flip):
    rng = np.random.RandomState(H * W + nw + flip)
    img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
    out = torch.zeros(S, S, 3, dtype=torch.uint8, device=dev)
    L.aug_place(torch.from_numpy(img).to(dev), False, out, S, nw, nh, dx, dy, flip)
    mask = (rng.rand(H, W) < 0.4).astype(np.uint8)
    mout = torch.zeros(S, S, dtype=torch.uint8, device=dev)
    L.aug_place(torch.from_numpy(mask).to(dev), True, mout, S, nw, nh, dx, dy, flip)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), O.place_image(img, S, nw, nh, dx, dy, flip))
    np.testing.assert_array_equal(mout.cpu().numpy().astype(bool), O.place_mask(mask, S, nw, nh, dx, dy, flip))
    f = torch.zeros(S, S, 3, dtype=torch.float32, device=dev)
    L.aug_to_float(out, f)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(f.cpu().numpy(), out.cpu().numpy().astype(np.float32) / 255.0)


def test_photometric_kernels_match_oracle(dev):
    rng = np.random.RandomState(11)
    S = 48
    img = rng.randint(0, 256, (S, S, 3)).astype(np.uint8)
    img[:4, :4] = 0
    img[4:8, :4] = 255
    img[8:12, :4] = [[[10, 10, 10]]]                                   # greys: S = 0 branch
    d = torch.from_numpy(img).to(dev)
    for coeff in (0.5, 0.77, 1.0, 1.31, 1.5):
        x = d.clone()
        L.aug_change_light(x, S, coeff)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(x.cpu().numpy(), O.change_light(img, coeff), err_msg="coeff %g" % coeff)
    for angle in (0, 45, 90, 135):
        for lt in (0, 1, 2):
            y = torch.zeros_like(d)
            L.aug_motion_blur3(d, y, S, angle, lt)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(y.cpu().numpy(), O.motion_blur3(img, angle, lt), err_msg="%d %d" % (angle, lt))
    ns, npp = 40, 90
    rows, cols = rng.randint(0, S - 1, ns + npp), rng.randint(0, S - 1, ns + npp)
    rows[ns], cols[ns] = rows[0], cols[0]                               # a pepper on a salt: pepper wins (applied second)
    x = d.clone()
    L.aug_salt_pepper(x, S, torch.from_numpy(rows.astype(np.int32)).to(dev), torch.from_numpy(cols.astype(np.int32)).to(dev), ns, npp)
    torch.cuda.synchronize()
    want = O.salt_pepper(img, rows, cols, ns)
    np.testing.assert_array_equal(x.cpu().numpy(), want)
    assert (want[rows[0], cols[0]] == 0).all()


def _labels(rng, n):
    out = []
    for k in range(n):
        h, w = rng.randint(120, 260), rng.randint(120, 260)
        image = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        polys, names = [], []
        for j in range(rng.randint(1, 4)):
            cy, cx = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w
            ry, rx = rng.uniform(0.08, 0.22) * h, rng.uniform(0.08, 0.22) * w
            t = np.sort(rng.uniform(0, 2 * np.pi, rng.randint(5, 9)))
            inst = [{"type": "out", "all_points_x": np.clip(cx + rx * np.cos(t), 0, w - 1).astype(int).tolist(),
                     "all_points_y": np.clip(cy + ry * np.sin(t), 0, h - 1).astype(int).tolist()}]
            if j == 0:
                inst.append({"type": "in", "all_points_x": [int(cx - 2), int(cx + 2), int(cx)], "all_points_y": [int(cy - 2), int(cy - 2), int(cy + 2)]})
            polys.append(inst)
            names.append(cfg.CLASSES[rng.randint(0, 3)])
        out.append({"image": image, "class_names": names, "polygons": polys})
    return out


class RecordingRandom(np.random.RandomState):
    def __init__(self, seed):
        super().__init__(seed)
        self.calls = []

    def randint(self, *a, **k):
        self.calls.append("randint")
        return super().randint(*a, **k)

    def uniform(self, *a, **k):
        self.calls.append("uniform")
        return super().uniform(*a, **k)

    def shuffle(self, x):
        self.calls.append("shuffle")
        return super().shuffle(x)


def test_defect_train_get_replayed_by_the_oracle(dev):
    S, B = 96, 4
    labels = _labels(np.random.RandomState(2), 5)
    rng = RecordingRandom(123)
    data = TD.defect_train(labels, batch_size=B, image_size=S, device=dev, rng=rng)
    seen = {"scale": set(), "flip": set(), "bnl": set()}
    for it in range(6):
        rng.calls.clear()
        order = [l for l in data.random_labels]
        cursor = data.cursor
        images, masks, tboxes, y3, y2, y1, window = data.get()
        torch.cuda.synchronize()
        assert images.shape == (B, S, S, 3) and masks.shape == (B, cfg.MAX_BOX_PER_IMAGE, S, S) and masks.dtype == torch.bool
        assert tboxes.shape == (B, 1, 1, 1, cfg.MAX_BOX_PER_IMAGE, 5) and y3.shape == (B, S // 8, S // 8, 3, 8)
        assert y1.shape == (B, S // 32, S // 32, 3, 8) and (window == [0, 0, 1, 1]).all()
        # the random draws come in the reference's order: scale/crop decision first (then its uniforms), flip, bnl
        assert rng.calls[0] == "randint"
        for b in range(B):
            lab = order[(cursor + b) % len(order)] if cursor + b < len(order) else None
            dec = data.last_decisions[b]
            seen["scale"].add(dec["scale_crop"]); seen["flip"].add(dec["flip"]); seen["bnl"].add(dec["bnl"])
            if lab is None:
                continue                                       # the epoch wrapped inside this batch: order reshuffled
            h, w = lab["image"].shape[:2]
            img = O.place_image(lab["image"], S, dec["new_w"], dec["new_h"], dec["dx"], dec["dy"], dec["flip"])
            if dec["bnl"] == 2:
                rows = np.concatenate([dec["salt"][0], dec["pepper"][0]]); cols = np.concatenate([dec["salt"][1], dec["pepper"][1]])
                img = O.salt_pepper(img, rows, cols, len(dec["salt"][0]))
            elif dec["bnl"] == 3:
                img = O.change_light(img, dec["coeff"])
            elif dec["bnl"] == 4:
                img = O.motion_blur3(img, dec["angle"], {"right": 1, "left": 2, "full": 0}[dec["line_type"]])
            np.testing.assert_array_equal(images[b].cpu().numpy(), img.astype(np.float32) / 255.0)
            n = len(lab["polygons"])
            for j in range(n):
                m = O.place_mask(O.instance_mask(lab["polygons"][j], h, w).astype(np.float32), S, dec["new_w"], dec["new_h"],
                                 dec["dx"], dec["dy"], dec["flip"])
                np.testing.assert_array_equal(masks[b, j].cpu().numpy(), m)
                # the box of the target arrays follows its mask (to the pixel rounding of the resize)
                xc, yc, bw, bh = tboxes[b, 0, 0, 0, j, :4] * S
                if m.any():
                    rows_, cols_ = np.where(m)
                    assert abs((cols_.min() + cols_.max() + 1) / 2 - xc) <= 2.5 and abs((rows_.min() + rows_.max() + 1) / 2 - yc) <= 2.5
                    assert abs((cols_.max() + 1 - cols_.min()) - bw) <= 3 and abs((rows_.max() + 1 - rows_.min()) - bh) <= 3
                assert tboxes[b, 0, 0, 0, j, 4] == cfg.CLASSES.index(lab["class_names"][j])
            assert not masks[b, n:].any() and (tboxes[b, 0, 0, 0, n:] == 0).all()
            # every target cell holds the box of one of the image's instances, in the cell its centre falls in
            for grid in (y3, y2, y1):
                gsz = grid.shape[1]
                for (yy, xx, a) in np.argwhere(grid[b, ..., 4] == 1):
                    row = grid[b, yy, xx, a]
                    assert int(row[0] * gsz) == xx and int(row[1] * gsz) == yy and row[5:].sum() == 1
                    assert any(np.allclose(row[:4], tboxes[b, 0, 0, 0, j, :4], atol=1e-6) for j in range(n))
            assert sum(int((g_[b, ..., 4] == 1).sum()) for g_ in (y3, y2, y1)) >= 1
    assert seen["scale"] == {1, 2} and seen["flip"] == {1, 2, 3} and seen["bnl"] == {1, 2, 3, 4}
    assert data.epoch >= 4                                      # 24 samples out of 5 labels: the cursor wrapped
    # same seed -> same batches
    a = TD.defect_train(labels, batch_size=B, image_size=S, device=dev, rng=np.random.RandomState(9)).get()
    b_ = TD.defect_train(labels, batch_size=B, image_size=S, device=dev, rng=np.random.RandomState(9)).get()
    assert torch.equal(a[0], b_[0]) and torch.equal(a[1], b_[1]) and np.array_equal(a[2], b_[2]) and np.array_equal(a[3], b_[3])


def test_defect_train_record_cache_and_buffer_sets(dev):
    """round 6: the per-record cache (image, instance masks, boxes on the GPU) changes nothing -- twelve batches with it against
    twelve without (cache_bytes=0: everything recomputed per visit, as before), same seed, bit for bit; and a batch stays intact
    through the next get() (two buffer sets), which is what a loop that reads one batch ahead holds on to."""
    S, B = 96, 2
    labels = _labels(np.random.RandomState(11), 5)
    a = TD.defect_train(labels, batch_size=B, image_size=S, device=dev, rng=np.random.RandomState(3))
    b = TD.defect_train(labels, batch_size=B, image_size=S, device=dev, rng=np.random.RandomState(3), cache_bytes=0)
    prev = None
    for _ in range(12):
        ga, gb = a.get(), b.get()
        assert not b._cache and len(a._cache) >= 1
        assert torch.equal(ga[0], gb[0]) and torch.equal(ga[1], gb[1])
        for x, y in zip(ga[2:], gb[2:]):
            assert np.array_equal(x, y)
        if prev is not None:
            assert torch.equal(prev[0], prev[2]) and torch.equal(prev[1], prev[3])     # the batch before this one: untouched
        prev = (ga[0], ga[1], ga[0].clone(), ga[1].clone())
    assert a.epoch == b.epoch and a.epoch >= 4


def test_defect_train_feeds_the_solver(dev, tmp_path):
    """end to end: polygons -> GPU pipeline -> Solver.train -> finite losses (the reference's main(), :237-248)"""
    from disyolo_amd.net import YOLONet
    from disyolo_amd.solver import Solver
    S, B = 64, 2
    data = TD.defect_train(_labels(np.random.RandomState(5), 4), batch_size=B, image_size=S, device=dev,
                           rng=np.random.RandomState(1))
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
    net.shuffle_seed = 1
    hist = Solver(net, data, output_dir=str(tmp_path), max_iter=6, summary_iter=3, save_iter=6, log=lambda s: None).train()
    assert len(hist) == 6 and np.isfinite(hist).sum() >= 4
