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


@pytest.mark.parametrize("H,Cin,Cout", [(36, 256, 512), (72, 128, 256), (18, 512, 1024)])
def test_conv_kernel_families_agree_at_full_size(dev, H, Cin, Cout):
    """the same 3x3 layer of the B=8 network through every kernel family -- GEMM tiles, split-K
    groups, the halo-reuse patch kernel -- with integer-valued operands whose products and sums
    are exact in f32: all summation orders must give the bit-identical f32 result, and the bf16
    epilogue (scale/shift, leaky, residual) the bit-identical bf16 result."""
    yb_probe = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device=dev)
    for tile in (19, 24, 25):     # the new families really run at these shapes (no silent fallback)
        assert L.conv2d_tile(L.make_conv_desc(yb_probe[..., :Cin].contiguous() if Cin <= Cout else torch.empty(B, H, H, Cin, dtype=torch.bfloat16, device=dev),
                                              torch.empty(Cout, 9 * Cin, dtype=torch.bfloat16, device=dev), yb_probe, 3, 1, tile=tile))[0] == tile
    g = torch.Generator(device=dev).manual_seed(H)
    x = torch.randint(-3, 4, (B, H, H, Cin), device=dev, generator=g).to(torch.bfloat16)
    w = torch.randint(-2, 3, (Cout, 9 * Cin), device=dev, generator=g).to(torch.bfloat16)
    res = torch.randint(-8, 9, (B, H, H, Cout), device=dev, generator=g).to(torch.bfloat16)
    scale = torch.full((Cout,), 2.0 ** -9, device=dev)
    shift = torch.full((Cout,), 0.25, device=dev)
    outs_f32, outs_bf16 = [], []
    # (19 = patch kernel with 16 channels per block; 24 / 25 = flat-frame kernels on 32x32x16 MFMA: bf16 output only, their f32
    #  launch falls back to the heuristic tile)
    for tile in (3, 6, 10, 12, 0x20d, 14, 16, 17, 18, 19, 24, 25):
        y = torch.empty(B, H, H, Cout, dtype=torch.float32, device=dev)
        L.conv2d_fwd(L.make_conv_desc(x, w, y, 3, 1, out_f32=True, tile=tile))
        yb = torch.empty(B, H, H, Cout, dtype=torch.bfloat16, device=dev)
        L.conv2d_fwd(L.make_conv_desc(x, w, yb, 3, 1, scale=scale, shift=shift, residual=res, leaky=True, tile=tile))
        outs_f32.append(y)
        outs_bf16.append(yb)
    torch.cuda.synchronize()
    assert float(outs_f32[0].abs().max()) > 100          # a real reduction, not zeros
    for y, yb in zip(outs_f32[1:], outs_bf16[1:]):
        assert torch.equal(y, outs_f32[0])
        assert torch.equal(yb, outs_bf16[0])


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


def test_config0_single_image_576_forward_matches_oracle(dev):
    """BASELINE.json configs[0]: 1x576x576, 3 classes, inference forward (the
    calculate_test_map.py path) -- the CPU oracle at full size against the HIP path:
    head logits, score maps, filtered detections and assembled masks."""
    import disyolo_oracle as O
    net = YOLONet(training=False, device=dev, image_size=S, batch_size=1, stage=1, seed=0)
    gen = torch.Generator().manual_seed(4242)
    with torch.no_grad():
        for i in (59, 67, 75, 82):
            net.params["yolo/convolutional%d/weights" % i].mul_(4.0)
            bias = net.params["yolo/convolutional%d/biases" % i]
            bias.copy_((torch.randn(bias.shape, generator=gen) * 0.3).to(bias.device))   # seeded: same logits every run
    net.refresh_weights()
    b = synthetic_batch(1, S, seed=123)
    window = np.array([[0.05, 0.0, 0.95, 1.0]], np.float32)            # letterbox-style clip window
    det_box, det_mask = net.evaluation(b["images"], window, [0.2])
    torch.cuda.synchronize()
    p = {k: v.detach().cpu() for k, v in net.params.items()}
    img = torch.from_numpy(b["images"])
    yq, mq = O.build_network(p, img, False, O.default_lock(1), quant=O.bf16_ste)
    yf, mf = O.build_network(p, img, False, O.default_lock(1))
    preds = [net.by_idx[i].act.cpu().view(1, net.by_idx[i].Ho, net.by_idx[i].Wo, 3, 8) for i in (75, 67, 59)]
    for got, wq, wf in list(zip(preds, yq, yf)) + [(net.by_idx[82].act.cpu(), mq, mf)]:
        e_q = float((got.double() - wq.double()).norm() / wq.double().norm())
        e_f = float((got.double() - wf.double()).norm() / wf.double().norm())
        assert e_q < 2e-2 and e_f < 8e-2, (e_q, e_f)
    # detection filter + mask assembly: exact on the kernels' own logits / score maps
    pred = O.interpret_output(preds)
    want_det = O.filter_detections(pred[2], pred[3], pred[5], window, 0.2)
    np.testing.assert_allclose(net.detections.cpu().numpy(), want_det, rtol=1e-5, atol=1e-6)
    wb, wm = O.val_test(net.detections.cpu().numpy(), net.by_idx[82].act.cpu())
    assert len(wb[0]) > 0
    np.testing.assert_array_equal(det_box[0], wb[0])
    np.testing.assert_allclose(det_mask[0], wm[0], rtol=1e-5, atol=1e-6)
    assert (wb[0][:, 0] >= 0.05 - 1e-6).all() and (wb[0][:, 2] <= 0.95 + 1e-6).all()


@pytest.mark.parametrize("size,batch", [(576, 2), (832, 1), (96, 3), (160, 2)])
def test_fused_launches_equal_the_layer_by_layer_forward_at_full_size(dev, size, batch, monkeypatch):
    """Size-independent property of the fused launches (conv1+2, residual blocks 1-3, mask head): at the BASELINE sizes the
    inference forward with them equals the layer-by-layer forward -- the same products, f32 sums in another order, each
    intermediate rounded to bf16 where the unfused path stores it: the head logits and score maps agree well inside the
    tolerance either path has against the oracle, and the plan really replaces the eleven layers."""
    def run(fused):
        monkeypatch.setenv("DISYOLO_FUSE_B64", "1" if fused else "0")
        net = YOLONet(training=False, device=dev, image_size=size, batch_size=batch, stage=1, seed=3)
        net.fuse_first_two = net.fuse_blocks = fused
        plan = net._fusion_plan(False, 1, 82)
        b = synthetic_batch(batch, size, seed=77)
        preds, det, mask_pos = net.forward(b["images"], b["clip_window"], [0.3], is_training=False)
        torch.cuda.synchronize()
        return sorted(plan), [t.float().cpu().clone() for t in preds] + [mask_pos.float().cpu().clone()]
    plan_f, out_f = run(True)
    plan_u, out_u = run(False)
    # (the 144^2-type blocks need a map width that is a multiple of 16: not at 96^2 / 160^2 inputs, which then run them
    #  layer by layer -- the plan is per size)
    assert plan_f == [1, 2, 3, 4] + ([6, 7, 8, 9] if (size // 4) % 16 == 0 else []) + [80, 81, 82] and plan_u == []
    for a, b_ in zip(out_f, out_u):
        assert torch.isfinite(a).all()
        scale = float(b_.abs().max())
        # (conv1 on split bf16 operands differs from the exact-f32 kernel by 2^-16 per product: ~3 % of act1's bf16 roundings
        #  flip, and 80 layers carry that to 0.7 % of the outputs' norm -- a third of what either path is from the oracle)
        assert float((a - b_).abs().max()) <= 2.0 ** -4 * scale, (float((a - b_).abs().max()), scale)
        assert float((a - b_).norm() / b_.norm()) < 1.5e-2


def test_empty_and_saturated_detection_edge_cases(dev):
    """no candidate above the threshold -> zero rows, count 0, evaluation returns the scalar
    0.0 mask (yolo/yolo3_net_pos.py:933); more than 30 survivors -> exactly 30, best first."""
    net = YOLONet(training=False, device=dev, image_size=192, batch_size=2, stage=1, seed=0)
    b = synthetic_batch(2, 192, seed=1)
    box, mask = net.evaluation(b["images"], b["clip_window"], [0.999])
    assert net.det_count.cpu().tolist() == [0, 0] and float(net.detections.abs().sum()) == 0.0
    assert all(bx.shape == (0, 6) for bx in box) and all(np.ndim(m) == 0 and m == 0.0 for m in mask)
    # flood: huge positive confidence everywhere -> every candidate passes; NMS still caps at 30
    with torch.no_grad():
        for i in (59, 67, 75):
            net.params["yolo/convolutional%d/biases" % i].view(3, 8)[:, 4] = 8.0
            net.params["yolo/convolutional%d/biases" % i].view(3, 8)[:, 2:4] = -2.5   # small boxes: little overlap
    net.refresh_weights()
    box, mask = net.evaluation(b["images"], b["clip_window"], [0.01])
    assert net.det_count.cpu().tolist() == [30, 30]
    d = net.detections.cpu().numpy()
    assert (np.diff(d[:, :, 5], axis=1) <= 0).all() and (d[:, :, 5] > 0.01).all()


def test_first_layer_matrix_core_kernel_equals_the_vector_kernel_at_full_size(dev):
    """Layer 1 at B=2, 576x576: the f32-MFMA kernel (sums the 27 taps in pairs) against the thread-per-pixel
    kernel (one sequential FMA chain), same f32 image and weights: both are exact-f32 products, the only
    difference is the summation order of 27 terms, so after the rounding to bf16 they agree bit for bit nearly
    everywhere and within one bf16 ulp elsewhere."""
    import os
    import subprocess
    import sys
    g = torch.Generator().manual_seed(9)
    B, S = 2, 576
    img = torch.rand(B, S, S, 3, generator=g).to(dev)
    w = (torch.randn(3, 3, 3, 32, generator=g) * 0.2).to(dev)
    sc = (torch.rand(32, generator=g) + 0.5).to(dev)
    sh = (torch.randn(32, generator=g) * 0.1).to(dev)
    y = torch.empty(B, S, S, 32, dtype=torch.bfloat16, device=dev)
    L.conv_first_fwd(img, w, sc, sh, y, alpha=0.1)
    torch.cuda.synchronize()
    # the vector kernel is selected by an environment variable read once per process: run it in a child
    path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out")
    os.makedirs(path, exist_ok=True)
    f_in, f_out = os.path.join(path, "first_in.pt"), os.path.join(path, "first_out.pt")
    torch.save({"img": img.cpu(), "w": w.cpu(), "sc": sc.cpu(), "sh": sh.cpu()}, f_in)
    code = ("import sys, torch; sys.path.insert(0, %r); import disyolo_amd; from disyolo_amd import lib as L; "
            "d = torch.load(%r); dev = torch.device('cuda:0'); "
            "y = torch.empty(%d, %d, %d, 32, dtype=torch.bfloat16, device=dev); "
            "L.conv_first_fwd(d['img'].to(dev), d['w'].to(dev), d['sc'].to(dev), d['sh'].to(dev), y, alpha=0.1); "
            "torch.cuda.synchronize(); torch.save(y.cpu(), %r)") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), f_in, B, S, S, f_out)
    env = dict(os.environ, DISYOLO_FIRST_VALU="1")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=600)
    want = torch.load(f_out).float()
    got = y.float().cpu()
    same = float((got == want).float().mean())
    assert same > 0.999, same
    ulp = 2.0 ** -7 * want.abs().clamp_min(2.0 ** -20)
    assert bool(((got - want).abs() <= ulp).all())
    os.remove(f_in), os.remove(f_out)
