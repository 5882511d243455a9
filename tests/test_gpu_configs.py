"""The BASELINE.json configurations that round 1 never ran on the GPU:

  * configs[1] stage 2 (all 82 layers trainable) at B=8 / 576x576: determinism, finiteness, and
    teacher-forced slices of the backward pass against a CPU f32 reference at FULL size;
  * configs[3] inference at B=32 through the hipGraph replay;
  * configs[2] the data-parallel step: the RCCL path with one rank must reproduce the plain step
    bit for bit (all-reduce of one rank is the identity; grad_scale = 1), with and without the
    cross-step backbone pipeline -- this runs the bucket cut list, the side-lane stream wrapping
    and finish()/Adam ordering of YOLONet.run_program on real hardware;
  * the effect of bf16 gradient storage along the residual trunk on conv1-10's weight gradients.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import disyolo_oracle as O
from disyolo_amd import config as cfg
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def seeded_heads(net, seed, gain=4.0, bias_std=0.3):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for i in (59, 67, 75, 82):
            net.params["yolo/convolutional%d/weights" % i].mul_(gain)
            b = net.params["yolo/convolutional%d/biases" % i]
            b.copy_((torch.randn(b.shape, generator=g) * bias_std).to(b.device))
    net.refresh_weights()


def rel_l2(got, want):
    got, want = got.double().flatten(), want.double().flatten()
    return float((got - want).norm() / (want.norm() + 1e-30))


def cpu_wgrad(x_nhwc, dy_nhwc, k, stride, cout):
    """dW (HWIO) of the TF-SAME convolution from NHWC activations, f32 on the host"""
    x = x_nhwc.float().permute(0, 3, 1, 2).contiguous()
    dy = dy_nhwc.float()[..., :cout].permute(0, 3, 1, 2).contiguous()
    H = x.shape[2]
    _, pb, pa = O.same_pads(H, k, stride)
    x = F.pad(x, (pb, pa, pb, pa))
    w = torch.nn.grad.conv2d_weight(x, (cout, x.shape[1], k, k), dy, stride=stride, padding=0)
    return w.permute(2, 3, 1, 0).contiguous()          # OIHW -> HWIO


def cpu_dgrad(dy_nhwc, w_hwio, k, stride, in_hw):
    dy = dy_nhwc.float().permute(0, 3, 1, 2).contiguous()
    w = w_hwio.float().permute(3, 2, 0, 1).contiguous()
    _, pb, pa = O.same_pads(in_hw, k, stride)
    gx = torch.nn.grad.conv2d_input((dy.shape[0], w.shape[1], in_hw + pb + pa, in_hw + pb + pa), w, dy, stride=stride, padding=0)
    gx = gx[:, :, pb:pb + in_hw, pb:pb + in_hw]
    return gx.permute(0, 2, 3, 1).contiguous()


@pytest.fixture(scope="module")
def stage2(dev):
    B, S = 8, 576
    batch = synthetic_batch(B, S, seed=91)
    rng = np.random.RandomState(3)
    batch["perm_det"] = np.stack([rng.permutation(cfg.MAX_DETECTION) for _ in range(B)]).astype(np.int32)
    batch["perm_gt"] = np.stack([rng.permutation(cfg.MAX_BOX_PER_IMAGE) for _ in range(B)]).astype(np.int32)
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=2, seed=5)
    seeded_heads(net, 11)
    net.set_batch(batch)
    return net, batch


def test_stage2_full_size_backward_is_deterministic_and_finite(stage2):
    net, _ = stage2
    grads = []
    for _ in range(2):
        net.grad_arena.zero_()
        net.compute_losses(0.2)
        net.backward()
        torch.cuda.synchronize()
        grads.append(net.grad_arena.clone())
    assert torch.equal(grads[0], grads[1])                       # fixed-order reductions everywhere
    assert bool(torch.isfinite(grads[0]).all())
    assert np.isfinite(float(net.total_loss().cpu())) and int(net.roi_count.sum()) > 0
    # every trainable variable received a gradient
    for name, (o, c) in net.arena_slices.items():
        assert float(grads[0][o:o + c].abs().max()) > 0, name


@pytest.mark.parametrize("idx", [1, 2, 4, 5, 12, 27, 44, 52])
def test_stage2_full_size_weight_gradient_slices_match_cpu(stage2, idx):
    """teacher forcing at full size: dW of layer idx from the HIP path's own input activation and output
    gradient against an f32 host convolution-gradient (layers chosen to cover the first-layer path, the
    stride-2 im2col kernel, the tap-fused 3x3 kernel at every ring size, and a 1x1 of the trunk)."""
    net, _ = stage2
    net.grad_arena.zero_()
    net.compute_losses(0.2)
    net.backward()
    torch.cuda.synchronize()
    l = net.by_idx[idx]
    got = l.dw.cpu()
    if idx == 1:
        # the first layer's weight gradient runs on the matrix cores over a bf16 copy of the f32 image
        # (forward uses the exact f32 image): compare with that arithmetic, and bound the distance to
        # the f32-image gradient by the bf16 input rounding (2^-9 relative per product)
        want = cpu_wgrad(net.images.to(torch.bfloat16).cpu(), l.dx.cpu(), l.k, l.stride, l.cout)
        exact = cpu_wgrad(net.images.cpu(), l.dx.cpu(), l.k, l.stride, l.cout)
        assert rel_l2(got, exact) < 6e-3, "layer 1 vs f32 image: rel l2 %.3g" % rel_l2(got, exact)
    else:
        want = cpu_wgrad(net.by_idx[l.src].act.cpu(), l.dx.cpu(), l.k, l.stride, l.cout)
    # f32 accumulation over up to 2.6 M pixels in a different order
    assert rel_l2(got, want) < 2e-3, "layer %d: rel l2 %.3g" % (idx, rel_l2(got, want))


@pytest.mark.parametrize("idx", [2, 4, 29])
def test_stage2_full_size_data_gradient_slices_match_cpu(stage2, idx):
    """the data gradient of layer idx (3x3 stride 1, stride 2, and a deep 3x3) from its dx: for a layer
    whose source has one consumer the source's grad buffer holds exactly conv_transpose(dx, w)"""
    net, _ = stage2
    net.grad_arena.zero_()
    net.compute_losses(0.2)
    net.backward()
    torch.cuda.synchronize()
    l = net.by_idx[idx]
    src = net.by_idx[l.src]
    consumers = [m.idx for m in net.layers if idx != m.idx and (m.src == src.idx or m.src_up == src.idx or m.shortcut == src.idx)]
    assert not consumers, "pick a layer whose input has a single consumer"
    w = net.params["yolo/convolutional%d/weights" % idx].cpu().to(torch.bfloat16).float()
    want = cpu_dgrad(l.dx.cpu(), w, l.k, l.stride, l.H)
    assert rel_l2(src.grad.float().cpu(), want) < 6e-3           # bf16 output rounding


def test_config3_inference_graph_at_batch_32(dev):
    """B=32, 576x576, hipGraph replay (network + detection filter + mask assembly).  The batch is 4
    distinct images repeated 8 times: rows of the implicit GEMM are independent, so identical images
    must give bit-identical detections and masks wherever they sit in the batch; the replay must equal
    the eager pass bit for bit; masks are 0.5 outside their boxes."""
    B, S = 32, 576
    net = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0)
    seeded_heads(net, 21)
    b4 = synthetic_batch(4, S, seed=9)
    images = np.concatenate([b4["images"]] * 8, axis=0)
    window = np.concatenate([b4["clip_window"]] * 8, axis=0)
    box_e, mask_e = net.evaluation(images, window, [0.2], masks_on_device=True)
    torch.cuda.synchronize()
    det_e, keep_e, masks_e = net.detections.clone(), net.keep.clone(), net.masks.clone()
    net.build_infer_program(det_thresh=0.2, graph=True)
    assert net._infer_graph is not None
    for _ in range(2):
        det, cnt, masks, keep = net.infer()
    torch.cuda.synchronize()
    assert torch.equal(det, det_e) and torch.equal(keep, keep_e) and torch.equal(masks, masks_e)
    assert int(cnt.sum()) > 0
    det, keep, masks = det.cpu().numpy(), keep.cpu().numpy().astype(bool), masks.cpu().numpy()
    for i in range(4, B):
        np.testing.assert_array_equal(det[i], det[i % 4])
        np.testing.assert_array_equal(keep[i], keep[i % 4])
        np.testing.assert_array_equal(masks[i][keep[i]], masks[i % 4][keep[i % 4]])
    Sm = S // 2
    for i in range(4):
        for r in np.where(keep[i])[0]:
            y1, x1, y2, x2 = (int(v) for v in np.round(det[i, r, :4] * np.float32(Sm)))
            outside = np.ones((Sm, Sm), bool)
            outside[y1:y2, x1:x2] = False
            assert (masks[i, r][outside] == 0.5).all()


@pytest.fixture(scope="module")
def one_rank_rccl(dev):
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dist
    # (a communicator torn down while device work is still queued is a classic way to hang: drain first)
    torch.cuda.synchronize()
    dist.destroy_process_group()


@pytest.mark.parametrize("inlist", [True, False])
@pytest.mark.parametrize("pipeline", [False, True])
def test_config2_one_rank_rccl_step_equals_plain_step(dev, one_rank_rccl, pipeline, inlist):
    """inlist: the gradient exchange as commands of the recorded step (csrc/comm.hip, the default on RCCL groups);
    False: the list cut at the bucket boundaries, torch.distributed issuing the collectives"""
    from disyolo_amd.dp import enable_data_parallel
    B, S = 2, 64
    batches = [O.synthetic_batch(B, S, seed=60 + t) for t in range(4)]
    nets = [YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=3) for _ in range(3)]
    for n in nets:
        seeded_heads(n, 31, gain=6.0, bias_std=0.5)
        n.shuffle_seed = 17
    dp_eager, dp_prog, plain = nets
    enable_data_parallel(dp_eager, bucket_mb=4.0, inlist=inlist)
    enable_data_parallel(dp_prog, bucket_mb=4.0, inlist=inlist)
    assert dp_prog.dp.inlist == inlist and (dp_prog.dp.comm is not None) == inlist
    assert len(dp_prog.dp.buckets) >= 3                      # several cuts of the recorded step (cut form)
    plain.build_program(det_thresh=0.1, pipeline_backbone=pipeline)
    dp_prog.build_program(det_thresh=0.1, pipeline_backbone=pipeline)
    if pipeline:
        for n in (plain, dp_prog):
            n._set_inputs(batches[0]["images"], batches[0]["clip_window"])
            n.prime_pipeline()
    losses = [[], [], []]
    for t in range(3):
        feed = dict(batches[t])
        dp_eager.set_batch(feed)
        losses[0].append(float(dp_eager.train_step(None, det_thresh=0.1).cpu()))
        if pipeline:
            feed = dict(batches[t])
            feed["images"] = batches[t + 1]["images"]        # labels of batch t, images of batch t+1
        for k, n in ((1, dp_prog), (2, plain)):
            n.set_batch(feed)
            losses[k].append(float(n.train_step(None).cpu()))
    torch.cuda.synchronize()
    assert losses[0] == losses[1] == losses[2]
    assert (len(dp_prog._prog_marks) == 0) == inlist         # the in-list exchange leaves the list uncut
    for n in (dp_eager, dp_prog):
        assert torch.equal(n.arena, plain.arena) and torch.equal(n.adam_m, plain.adam_m) and torch.equal(n.adam_v, plain.adam_v)
        for name in plain.params:
            assert torch.equal(n.params[name], plain.params[name]), name


@pytest.mark.parametrize("sync_bn", [False, True])
def test_config2_exchange_in_the_list_with_the_overlapped_tail(dev, one_rank_rccl, sync_bn):
    """the data-parallel step as bench.py runs it since round 5: collectives recorded on the list's exchange lane, each
    slice's optimizer sweep right behind its all-reduce, the tail left open into the next replay (overlap_tail) --
    against the plain joined single-GPU step on the same batches, bit for bit; un-joined replays in between.  With
    SyncBN the per-layer statistics all-reduces are commands of the lane that produced the sums."""
    from disyolo_amd.dp import enable_data_parallel
    B, S = 2, 64
    batches = [O.synthetic_batch(B, S, seed=80 + t) for t in range(5)]
    nets = [YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=3) for _ in range(2)]
    for n in nets:
        seeded_heads(n, 31, gain=6.0, bias_std=0.5)
        n.shuffle_seed = 17
        n.set_batch(batches[0])
    plain, dpn = nets
    enable_data_parallel(dpn, inlist=True, sync_bn=sync_bn)
    plain.build_program(det_thresh=0.1)
    dpn.build_program(det_thresh=0.1, overlap_tail=True)
    assert dpn._prog_marks == [] and dpn._overlap
    for t in range(5):
        plain.set_batch(batches[t])
        dpn.set_batch(batches[t])
        assert t == 0 or dpn._tail_open
        plain.train_step(None, want_loss=False)
        dpn.train_step(None, want_loss=False)
    l0, l1 = float(plain.total_loss().cpu()), float(dpn.total_loss().cpu())
    torch.cuda.synchronize()
    assert l0 == l1
    assert torch.equal(dpn.arena, plain.arena) and torch.equal(dpn.adam_m, plain.adam_m) and torch.equal(dpn.adam_v, plain.adam_v)
    for name in plain.params:
        assert torch.equal(dpn.params[name], plain.params[name]), name
    d = dpn.dp.describe()
    assert d["list_cuts"] == 0 and d["collectives_per_step"] == len(dpn.opt_chunks) + 1


@pytest.mark.parametrize("inlist", [True, False])
@pytest.mark.parametrize("wire,algo", [("f32", "rs_ag"), ("bf16", "allreduce"), ("bf16", "rs_ag")])
def test_config2_wire_formats_and_reduce_scatter_variant(dev, one_rank_rccl, wire, algo, inlist):
    """the exchange options of dp.py on one RCCL rank: reduce-scatter + all-gather of an f32 bucket is the
    identity (bit-identical step); the bf16 wire rounds the gradient to 8 significant bits once, so after
    one Adam step (|dw| <= lr) the weights agree to a fraction of lr"""
    from disyolo_amd.dp import enable_data_parallel
    B, S = 2, 64
    batch = O.synthetic_batch(B, S, seed=71)
    nets = [YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=1, seed=3) for _ in range(2)]
    for n in nets:
        seeded_heads(n, 31, gain=6.0, bias_std=0.5)
        n.shuffle_seed = 5
        n.set_batch(batch)
    plain, dpn = nets
    enable_data_parallel(dpn, bucket_mb=4.0, wire=wire, algo=algo, inlist=inlist)
    plain.build_program(det_thresh=0.1)
    dpn.build_program(det_thresh=0.1)
    # bf16 wire: ONE step (from identical weights); the rounded gradients change the weights in the last
    # bits, and a second forward pass may then make a different discrete decision (NMS survivor, RoI set)
    for _ in range(2 if wire == "f32" else 1):
        l0 = float(plain.train_step(None).cpu())
        l1 = float(dpn.train_step(None).cpu())
        assert l0 == l1
    torch.cuda.synchronize()
    if wire == "f32":
        assert torch.equal(plain.arena, dpn.arena) and torch.equal(plain.adam_v, dpn.adam_v)
    else:
        dw = (plain.arena - dpn.arena).abs()
        i = int(dw.argmax())
        info = "max |dw| %.3g at %d: grads %.6g / %.6g, rel l2 of the gradients %.3g" % (
            float(dw.max()), i, float(plain.grad_arena[i]), float(dpn.grad_arena[i]), rel_l2(dpn.grad_arena, plain.grad_arena))
        assert rel_l2(dpn.grad_arena, plain.grad_arena) < 2 ** -8, info
        assert float(dw.max()) < 0.5 * cfg.LEARNING_RATE, info


@pytest.mark.parametrize("inlist", [True, False])
def test_sync_bn_on_one_rank_is_the_plain_step(dev, one_rank_rccl, inlist):
    """SyncBN with a single rank: the all-reduces are identities, the statistics go through the split
    (partial rows -> f64 sums -> finalize) kernels instead of the fused ones -- same arithmetic, so the
    recorded step must reproduce the plain step bit for bit (eager and recorded)."""
    from disyolo_amd.dp import enable_data_parallel
    B, S = 2, 64
    batch = O.synthetic_batch(B, S, seed=72)
    nets = [YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=2, seed=3) for _ in range(3)]
    for n in nets:
        seeded_heads(n, 31, gain=6.0, bias_std=0.5)
        n.shuffle_seed = 5
        n.set_batch(batch)
    plain, sync_eager, sync_prog = nets
    enable_data_parallel(sync_eager, bucket_mb=4.0, sync_bn=True, inlist=inlist)
    enable_data_parallel(sync_prog, bucket_mb=4.0, sync_bn=True, inlist=inlist)
    plain.build_program(det_thresh=0.1)
    sync_prog.build_program(det_thresh=0.1)
    # cut form: the recorded list is cut at the statistics; in-list form: their all-reduces are commands, no cut
    assert any(isinstance(w, tuple) for _, w in sync_prog._prog_marks) != inlist
    for _ in range(2):
        ls = [float(n.train_step(None, det_thresh=0.1).cpu()) for n in (plain, sync_eager, sync_prog)]
        assert ls[0] == ls[1] == ls[2]
    torch.cuda.synchronize()
    for n in (sync_eager, sync_prog):
        assert torch.equal(n.arena, plain.arena) and torch.equal(n.adam_v, plain.adam_v)
        for name in plain.params:
            assert torch.equal(n.params[name], plain.params[name]), name


def test_bf16_gradient_storage_along_the_residual_trunk(dev):
    """Stage 2, 192x192, B=2: weight gradients of conv1-10 (the far end of 23 residual blocks whose
    trunk gradient is rounded to bf16 once per block) and of conv43-52 (the near end) against the oracle
    differentiated in f32 at the HIP path's own activations.  Records the numbers."""
    B, S = 2, 192
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=2, seed=2)
    seeded_heads(net, 41, gain=6.0, bias_std=0.5)
    b = O.synthetic_batch(B, S, seed=13)
    rng = np.random.RandomState(1)
    perms = [(rng.permutation(cfg.MAX_DETECTION).astype(np.int32), rng.permutation(cfg.MAX_BOX_PER_IMAGE).astype(np.int32))
             for _ in range(B)]
    b["perm_det"], b["perm_gt"] = np.stack([p[0] for p in perms]), np.stack([p[1] for p in perms])
    p0 = {k: v.detach().cpu().float().clone() for k, v in net.params.items()}
    lock = O.default_lock(2)
    net.set_batch(b)
    net.compute_losses(0.1)
    net.backward()
    torch.cuda.synchronize()
    tr = {n: p0[n].clone().requires_grad_(True) for n in O.trainable_names(lock)}
    pp = dict(p0)
    pp.update(tr)
    force = {"act%d" % l.idx: l.act.float().cpu() for l in net.layers}
    parts, _, _, _ = O.total_loss(pp, b, lock, True, perms, {}, obj_thresh=0.1, quant=O.bf16_ste, taps={}, force=force)
    parts["total"].backward()
    errs = {}
    for i in list(range(1, 11)) + list(range(43, 53)):
        name = "yolo/convolutional%d/weights" % i
        o, c = net.arena_slices[name]
        want = tr[name].grad.flatten() - O.L2_WEIGHT * tr[name].detach().flatten()
        errs[i] = rel_l2(net.grad_arena[o:o + c].cpu(), want)
    far, near = [errs[i] for i in range(1, 11)], [errs[i] for i in range(43, 53)]
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "resid_grad_errors.json"), "w") as f:
        json.dump({"rel_l2_error_by_layer": errs, "median_conv1_10": float(np.median(far)),
                   "median_conv43_52": float(np.median(near))}, f, indent=1)
    assert max(far + near) < 0.05, errs
