"""The two callers of the hot path: ``evaluate`` (calculate_test_map.py:180-347) and
``Solver.train`` (train_yolo3_mask.py:117-227), plus ``image_read`` (calculate_test_map.py:149-176)."""
import json
import os

import numpy as np
import pytest
import torch

import disyolo_oracle as O
from disyolo_amd import checkpoint as ck
from disyolo_amd import config as cfg
from disyolo_amd import evaluate as E
from disyolo_amd.net import YOLONet
from disyolo_amd.solver import Solver, scheduled_learning_rate
from disyolo_amd.synth import synthetic_batch
from disyolo_amd.voc_eval import voc_eval

pytestmark = pytest.mark.gpu


def seeded_heads(net, seed, gain=6.0, bias_std=0.5):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for i in (59, 67, 75, 82):
            net.params["yolo/convolutional%d/weights" % i].mul_(gain)
            b = net.params["yolo/convolutional%d/biases" % i]
            b.copy_((torch.randn(b.shape, generator=g) * bias_std).to(b.device))
    net.refresh_weights()


@pytest.mark.parametrize("hw", [(348, 620), (600, 800), (754, 1008), (450, 386), (96, 96), (5, 300)])
@pytest.mark.parametrize("size", [576, 96])
def test_image_read_matches_oracle_bit_for_bit(dev, hw, size):
    """the four sample image sizes of data/train_sample plus a square and a sliver"""
    rng = np.random.RandomState(hw[0] + size)
    rgb = rng.randint(0, 256, size=(hw[0], hw[1], 3)).astype(np.uint8)
    got, window = E.image_read(rgb, size, dev)
    want, wwin = O.image_read(rgb, size)
    np.testing.assert_array_equal(window, wwin)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    top, left = int(round(float(window[0]) * size)), int(round(float(window[1]) * size))
    if top > 0:
        assert (got[:top].cpu().numpy() == np.float32(127.0 / 255.0)).all()
    if left > 0:
        assert (got[:, :left].cpu().numpy() == np.float32(127.0 / 255.0)).all()


def _fixture_set(S, seed):
    """three images of different sizes with elliptical ground-truth instances: (images, MAP)"""
    rng = np.random.RandomState(seed)
    images, recs, sizes, merged, index = {}, {}, {}, {}, []
    for k, (h, w) in enumerate([(150, 260), (200, 200), (240, 130)]):
        name = "img%03d" % k
        index.append(name)
        images[name] = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        sizes[name] = [h, w]
        yy, xx = np.mgrid[0:h, 0:w]
        objs, mm = [], np.zeros((h, w), np.uint8)
        for j in range(3):
            cy, cx, ry, rx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w, rng.uniform(0.1, 0.3) * h, rng.uniform(0.1, 0.3) * w
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            c = int(rng.randint(0, 3))
            objs.append({"imageid": name, "classid": c, "difficult": 0, "mask": m})
            mm[m] = c + 1
        recs[name], merged[name] = objs, mm
    return images, E.MAP(recs, sizes, index, merged, net_size=S)


def test_evaluate_matches_the_oracle_loop(dev):
    S = 192
    net = YOLONet(training=False, device=dev, image_size=S, batch_size=1, stage=1, seed=0)
    seeded_heads(net, 5)
    images, emap = _fixture_set(S, 3)
    thresh_out, mask_acc, timing = E.evaluate(net, images, emap, det_thresh=0.05)
    # the oracle loop on the HIP path's own detections / assembled masks (teacher forcing per image)
    detfile = {str(c): [] for c in range(3)}
    pred_maps, ndet = [], 0
    for name in emap.index:
        img, window = O.image_read(images[name], S)
        det_box, det_mask = net.evaluation(img[None], window[None], [np.float32(0.05)])
        h, w = emap.sizes[name]
        entries, merged = O.paste_detections(det_box[0], det_mask[0], h, w, S)
        ndet += len(entries)
        for e in entries:
            detfile[str(e["classid"])].append({"imageid": name, "score": e["score"], "mask": e["mask"]})
        pred_maps.append(merged)
    assert ndet >= 5, "fixture needs detections"
    aps, recs, precs = [], [], []
    for c in range(3):
        r, p, a = voc_eval(detfile[str(c)], emap.recs_mask, emap.index, c, ovthresh=0.5) if detfile[str(c)] else (0.0, 0.0, 0.0)
        recs, precs, aps = recs + [r], precs + [p], aps + [a]
    assert thresh_out[0]["AP"] == aps
    np.testing.assert_allclose(thresh_out[0]["mAP"], [np.mean(recs), np.mean(precs), np.mean(aps)], rtol=0, atol=1e-15)
    want_acc = O.segmentation_miou([emap.merged[n] for n in emap.index], pred_maps)
    np.testing.assert_allclose(mask_acc, want_acc, rtol=0, atol=1e-15)
    assert timing["per_image_s"] > 0
    # do_python_eval (utils/validation_map.py:104-198) over the same detections gives the same table
    detdata = []
    for name in emap.index:
        img, window = E.image_read(images[name], S, dev)
        b, m = net.evaluation(img[None], window[None], [np.float32(0.05)], masks_on_device=True)
        detdata.append({"boxes": b[0], "masks": m[0], "imname": name})
    assert emap.do_python_eval(detdata)[0]["AP"] == aps


class _TrainData:
    """stand-in for utils/train_data.defect_train: get() -> the seven arrays in the reference's order"""

    def __init__(self, B, S):
        self.batch_size, self.image_size, self.epoch, self.t = B, S, 1, 0

    def get(self):
        b = synthetic_batch(self.batch_size, self.image_size, seed=500 + self.t)
        self.t += 1
        return b["images"], b["true_masks"], b["true_boxes"], b["yolo3"], b["yolo2"], b["yolo1"], b["clip_window"]


class _ValData:
    def __init__(self, images, emap, S, dev):
        self.items = [(n,) + tuple(E.image_read(images[n], S, dev)) for n in emap.index]

    def get(self):
        return (torch.stack([it[1] for it in self.items]), [it[0] for it in self.items],
                np.stack([it[2] for it in self.items]))


def test_solver_train_loop_cadence_checkpoints_and_schedule(dev, tmp_path):
    B, S = 2, 64
    images, emap = _fixture_set(S, 9)
    images["img003"] = images["img000"][::-1].copy()           # 4 validation images = 2 batches
    emap.index.append("img003")
    emap.sizes["img003"], emap.recs_mask["img003"] = emap.sizes["img000"], emap.recs_mask["img000"]
    nets = [YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=4) for _ in range(3)]
    for n in nets:
        seeded_heads(n, 8)
    nets[1].shuffle_seed = nets[2].shuffle_seed = 20190530   # what the Solver gives a net that has none
    logs = []
    solver = Solver(nets[0], _TrainData(B, S), emap, _ValData(images, emap, S, dev), output_dir=str(tmp_path / "out"),
                    max_iter=20, summary_iter=2, save_iter=10, log=logs.append)
    hist = solver.train()
    # the same steps by hand on a twin: the driver adds nothing to the arithmetic
    twin, data = nets[1], _TrainData(B, S)
    twin.learning_rate = 1e-4
    want = []
    for step in range(20):
        im, tm, tb, y3, y2, y1, win = data.get()
        feed = {"images": im, "true_masks": tm, "true_boxes": tb, "yolo3": y3, "yolo2": y2, "yolo1": y1, "clip_window": win}
        if step == 0:
            twin.set_batch(feed)
            twin.build_program(det_thresh=cfg.OBJ_THRESHOLD)
        want.append(float(twin.train_step(feed).cpu()))
    # (a step whose mask loss meets a zero-area positive RoI is NaN, as in the reference -- SURVEY B14)
    np.testing.assert_array_equal(hist, want)
    assert len(hist) == 20 and np.isfinite(hist).sum() >= 15
    ckdir = tmp_path / "out" / "checkpoint"
    for step in (10, 20):
        assert (ckdir / ("model.ckpt-%d.index" % step)).exists() and (tmp_path / "out" / "lossnp" / ("%dmap.npy" % step)).exists()
    assert ck.latest_checkpoint(str(ckdir)).endswith("model.ckpt-20")
    got = ck.load_checkpoint(str(ckdir / "model.ckpt-20"))
    assert len(got) == 398 and all(np.array_equal(got[k], v.cpu().numpy()) for k, v in nets[0].params.items())
    assert not np.array_equal(ck.load_checkpoint(str(ckdir / "model.ckpt-10"))["yolo/convolutional82/weights"],
                              got["yolo/convolutional82/weights"])
    events = [json.loads(ln) for ln in open(ckdir / "events.jsonl")]
    assert [e["step"] for e in events] == list(range(2, 21, 2))
    assert set(events[0]) == {"object_loss", "noobject_loss", "class_loss", "xy_loss", "wh_loss", "mask_loss", "total_loss", "step"}
    assert "IMAGE_SIZE: 576" in open(ckdir / "config.txt").read()
    val_map = np.load(tmp_path / "out" / "lossnp" / "20map.npy")
    assert val_map.shape == (800, 9) and val_map[0, 0] == 20 and (val_map[1:] == 0).all()
    assert any("mAP50" in s and "Learning rate: 0.001" in s for s in logs)      # the printed (not applied) schedule
    # "intended": the schedule is applied -- a different trajectory from step 1 on
    s2 = Solver(nets[2], _TrainData(B, S), output_dir=str(tmp_path / "out2"), lr_schedule="intended", max_iter=3,
                summary_iter=50, save_iter=50, log=lambda s: None)
    h2 = s2.train()
    assert nets[2].learning_rate == scheduled_learning_rate(3) == 1e-3
    np.testing.assert_array_equal(h2[0], hist[0])
    assert h2[1] != hist[1]
    # restart from the checkpoint the way Solver.__init__ does for stage 1 (include list)
    fresh = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=99)
    Solver(fresh, _TrainData(B, S), output_dir=str(tmp_path / "out3"), restore_weight=str(ckdir / "model.ckpt-20"), stage=1,
           max_iter=0, log=lambda s: None)
    assert torch.equal(fresh.params["yolo/convolutional60/weights"], nets[0].params["yolo/convolutional60/weights"])
    assert not torch.equal(fresh.params["yolo/convolutional80/weights"], nets[0].params["yolo/convolutional80/weights"])


def test_host_feeder_on_the_pipelined_step_trains_like_set_batch(dev):
    """the same on a net whose recorded step is the pipelined one: the feeder mixes labels of batch t with the images of batch
    t + 1 itself, primes the first backbone pass and moves the batch on the net's feed stream -- five host batches, the plain
    net's weights bit for bit"""
    from disyolo_amd.feed import HostFeeder
    B, S, N = 2, 64, 5
    nets = [YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=4) for _ in range(2)]
    for n in nets:
        seeded_heads(n, 8)
        n.shuffle_seed = 3
    batches = [O.synthetic_batch(B, S, seed=300 + t) for t in range(N)]
    plain, fed = nets
    plain.set_batch(batches[0])
    plain.build_program(det_thresh=0.1)
    fed.set_batch(batches[0])
    fed.build_program(det_thresh=0.1, pipeline_backbone=True)
    for b in batches:
        plain.set_batch(b)
        plain.train_step(None, want_loss=False)
    feeder = HostFeeder(fed)
    feeder.submit(batches[0])
    for t in range(N):
        if t + 1 < N:
            feeder.submit(batches[t + 1])
        feeder.step(want_loss=False)
    torch.cuda.synchronize()
    lp, lq = plain.step_losses(0, N), fed.step_losses(0, N)
    assert lp.view(np.int32).tolist() == lq.view(np.int32).tolist(), (lp, lq)
    same = lambda a, b: torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert same(plain.arena, fed.arena) and same(plain.adam_v, fed.adam_v)


def test_host_feeder_trains_like_set_batch(dev):
    """feed.HostFeeder (pinned host batch -> copy stream -> staging -> input buffers, one batch ahead) changes when
    the inputs arrive, not what is computed: three steps on three different host batches leave the same weights,
    bit for bit, as set_batch + train_step."""
    from disyolo_amd.feed import HostFeeder
    B, S = 2, 64
    nets = [YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=4) for _ in range(2)]
    for n in nets:
        seeded_heads(n, 8)
        n.shuffle_seed = 3
    batches = [O.synthetic_batch(B, S, seed=300 + t) for t in range(3)]
    plain, fed = nets
    for n in nets:
        n.set_batch(batches[0])
        n.build_program(det_thresh=0.1)
    want = []
    for b in batches:
        plain.set_batch(b)
        want.append(float(plain.train_step(None).cpu()))
    feeder = HostFeeder(fed)
    feeder.submit(batches[0])
    got = []
    for t in range(3):
        if t + 1 < 3:
            feeder.submit(batches[t + 1])
        got.append(float(feeder.step().cpu()))
    with pytest.raises(RuntimeError):
        feeder.step()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got, want)        # (a NaN step -- zero-area positive RoI, SURVEY B14 -- is NaN in both)
    assert torch.equal(plain.arena, fed.arena) and torch.equal(plain.adam_v, fed.adam_v)

