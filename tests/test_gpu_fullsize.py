"""Full-size (B=8, 576x576) checks through size-independent properties: the oracle cannot
run this size in seconds, so these tests assert invariants any correct implementation has."""
import numpy as np
import pytest
import torch

from disyolo_amd import config as cfg
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

pytestmark = pytest.mark.gpu
B, S = 8, 576


@pytest.fixture(scope="module")
def trained(dev):
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
    with torch.no_grad():
        for i in (59, 67, 75, 82):
            net.params["yolo/convolutional%d/weights" % i].mul_(4.0)
    net.refresh_weights()
    net.set_batch(synthetic_batch(B, S, seed=77))
    return net


def _iou(a, b):
    iy = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    ix = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iy * ix
    ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / ua if ua > 0 else 0.0


def test_detection_filter_invariants_at_full_size(trained):
    net = trained
    net.compute_losses(0.2)
    torch.cuda.synchronize()
    det = net.detections.cpu().numpy()
    cnt = net.det_count.cpu().numpy()
    assert det.shape == (B, cfg.MAX_DETECTION, 6) and cnt.sum() > 0
    for b in range(B):
        n = int(cnt[b])
        rows = det[b, :n]
        assert (det[b, n:] == 0).all()                               # zero padding
        assert (np.diff(rows[:, 5]) <= 0).all()                      # score-descending
        assert (rows[:, 5] > 0.2).all()                              # thresholded
        assert (rows[:, :4] >= 0).all() and (rows[:, :4] <= 1).all()  # clipped to the window
        assert set(np.unique(rows[:, 4])) <= {0.0, 1.0, 2.0}
        for i in range(n):                                           # per-class NMS: survivors overlap <= 0.3
            for j in range(i):
                if rows[i, 4] == rows[j, 4]:
                    assert _iou(rows[i], rows[j]) <= cfg.IOU_THRESHOLD + 1e-6


def test_losses_finite_and_gradients_match_finite_difference_of_the_bias(trained):
    """d(total)/d(bias of conv82) from the backward pass against a central difference of the
    mask loss through the full-size forward (the bias shifts every score map value)."""
    net = trained
    net.compute_losses(0.2)
    net.backward()
    torch.cuda.synchronize()
    vals = net.losses.cpu().numpy()
    assert np.isfinite(vals).all() and np.isfinite(float(net.mask_loss.cpu()[0]))
    assert int(net.roi_count.sum()) > 0
    o, c = net.arena_slices["yolo/convolutional82/biases"]
    g = net.grad_arena[o:o + c].cpu().numpy().copy()
    bias = net.params["yolo/convolutional82/biases"]
    eps = 0.05
    num = np.zeros(c)
    for k in range(c):
        vals2 = []
        for sgn in (+1, -1):
            bias[k] += sgn * eps
            net.compute_losses(0.2)
            vals2.append(float(net.mask_loss.cpu()[0]))
            bias[k] -= sgn * eps
        num[k] = (vals2[0] - vals2[1]) / (2 * eps)
    assert np.abs(num).max() > 0
    np.testing.assert_allclose(g, num, rtol=0.05, atol=2e-3 * np.abs(num).max())


def test_train_step_is_deterministic_at_full_size(dev):
    """two identically seeded nets, three recorded steps each: bit-identical state (all
    reductions are fixed-order; no float atomics anywhere on the path)."""
    outs = []
    for _ in range(2):
        net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=3)
        net.set_batch(synthetic_batch(B, S, seed=5))
        net.build_program()
        for _ in range(3):
            net.train_step(None, want_loss=False)
        torch.cuda.synchronize()
        outs.append((net.arena.clone(), net.adam_v.clone(), float(net.total_loss().cpu())))
        del net
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and np.isfinite(outs[0][2])


def test_conv_linearity_at_full_size(dev):
    """conv(a*x) == a*conv(x) for a power of two (exact in bf16/f32) on the largest layer shape"""
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, 288, 288, 32, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(64, 9 * 32, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    y1 = torch.empty(B, 288, 288, 64, dtype=torch.float32, device=dev)
    y2 = torch.empty_like(y1)
    L.conv2d_fwd(L.make_conv_desc(x, w, y1, 3, 1, out_f32=True))
    L.conv2d_fwd(L.make_conv_desc((x.float() * 4).to(torch.bfloat16), w, y2, 3, 1, out_f32=True))
    torch.cuda.synchronize()
    assert torch.equal(y2, y1 * 4)
    # zero-padding: a border pixel only sees in-image taps
    ones = torch.ones(1, 64, 64, 32, device=dev, dtype=torch.bfloat16)
    wk = torch.ones(32, 9 * 32, device=dev, dtype=torch.bfloat16)
    y = torch.empty(1, 64, 64, 32, dtype=torch.float32, device=dev)
    L.conv2d_fwd(L.make_conv_desc(ones, wk, y, 3, 1, out_f32=True))
    torch.cuda.synchronize()
    assert float(y[0, 0, 0, 0]) == 4 * 32 and float(y[0, 0, 5, 0]) == 6 * 32 and float(y[0, 5, 5, 0]) == 9 * 32


def test_inference_masks_are_half_outside_boxes_at_full_size(dev):
    net = YOLONet(training=False, device=dev, image_size=S, batch_size=4, stage=1, seed=0)
    with torch.no_grad():
        for i in (59, 67, 75, 82):
            net.params["yolo/convolutional%d/weights" % i].mul_(4.0)
    net.refresh_weights()
    b = synthetic_batch(4, S, seed=9)
    net._set_inputs(b["images"], b["clip_window"])
    net.build_infer_program(det_thresh=0.2, graph=True)
    det, cnt, masks, keep = net.infer()
    torch.cuda.synchronize()
    det, keep, masks = det.cpu().numpy(), keep.cpu().numpy(), masks.cpu().numpy()
    assert keep.sum() > 0
    Sm = S // 2
    for bi in range(4):
        for r in range(cfg.MAX_DETECTION):
            if not keep[bi, r]:
                continue
            y1, x1, y2, x2 = (int(v) for v in np.round(det[bi, r, :4] * np.float32(Sm)))
            m = masks[bi, r]
            outside = np.ones((Sm, Sm), bool)
            outside[y1:y2, x1:x2] = False
            assert (m[outside] == 0.5).all()
            assert ((m > 0) & (m < 1)).all()
    # replaying the graph gives the same answer
    det2, _, masks2, _ = net.infer()
    torch.cuda.synchronize()
    assert np.array_equal(det2.cpu().numpy(), det) and np.array_equal(masks2.cpu().numpy(), masks)
