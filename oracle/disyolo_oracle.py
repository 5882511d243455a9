"""CPU oracle for the DIS-YOLO hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a function-by-function CPU restatement (torch-CPU, f32 or f64) of
the reference's TensorFlow-1.x graph ``yolo/yolo3_net_pos.py`` and of the
optimizer line in ``train_yolo3_mask.py``.  Every function cites the reference
lines it follows (paths relative to the reference checkout).

PARITY UNPINNED: TensorFlow 1.x is not installable in the build container and
the reference ships no tests, golden vectors, weights or checkpoints (SURVEY.md
F1-F4), so this oracle could not be checked against outputs of the reference
itself.  TF-op semantics it relies on are listed in SURVEY.md Appendix B and are
marked [TF-sem] below.  What *is* pinned: hand-derived known answers in
``tests/test_oracle_kat.py``; and, by outputs of the reference's own code run in the
build container (``tools/make_golden.py`` -> ``tests/golden/``): ``voc_eval`` /
``voc_ap`` / mask overlaps and the config constants (importable modules), and -- taken
out of their un-importable files with ``ast`` and executed -- the YOLO target
assignment loop (``utils/train_data.py:134-178`` -> ``assign_targets``), the box
un-letterboxing (``calculate_test_map.py:121-138`` -> ``correct_yolo_boxes``) and the
mIoU block (``:303-346`` -> ``segmentation_miou``).  The network graph, the losses,
NMS, the mask assembly and Adam remain unpinned.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``dis-yolo_amd/``) never does.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------
# constants restated from yolo/config.py:21-72 (checked against the golden
# config.json generated from the reference module, tests/test_config.py)
# ---------------------------------------------------------------------------
CLASSES = ["crack", "spall", "rebar"]
ANCHORS = np.array([[31, 23], [62, 58], [143, 91], [213, 186], [61, 337], [194, 432],
                    [474, 248], [551, 93], [478, 454]], dtype=np.float32)
ALPHA = 0.1
K_MAP = 3
OBJECT_SCALE, NOOBJECT_SCALE, CLASS_SCALE, COORD_SCALE, MASK_SCALE = 2.0, 1.0, 1.0, 1.0, 5.0
IGNORE_THRESH = 0.5
OBJ_THRESHOLD = 0.25
IOU_THRESHOLD = 0.3
MAX_BOX_PER_IMAGE = 20
MAX_DETECTION = 30
BN_DECAY, BN_EPS = 0.997, 1e-5          # yolo/yolo3_net_pos.py:74-75
L2_WEIGHT = 1e-4                         # yolo/yolo3_net_pos.py:38
NUM_ANCHOR = 3


# ---------------------------------------------------------------------------
# topology: (index, cin, cout, ksize, stride, kind) restated from
# yolo/yolo3_net_pos.py:159-412.  kind: 'bn' = conv_bn, 'res' = res_conv_bn,
# 'lin' = conv with bias and no activation.
# ---------------------------------------------------------------------------
def layer_specs() -> Dict[int, Tuple[int, int, int, int, str]]:
    s: Dict[int, Tuple[int, int, int, int, str]] = {}
    s[1] = (3, 32, 3, 1, "bn")                      # :159
    s[2] = (32, 64, 3, 2, "bn")                     # :165
    s[3] = (64, 32, 1, 1, "bn")                     # :170
    s[4] = (32, 64, 3, 1, "res")                    # :174
    s[5] = (64, 128, 3, 2, "bn")                    # :180
    for i in (6, 8):                                # :185,:194
        s[i] = (128, 64, 1, 1, "bn")
        s[i + 1] = (64, 128, 3, 1, "res")           # :189,:198
    s[10] = (128, 256, 3, 2, "bn")                  # :204
    for i in range(8):                              # :208-218
        s[2 * i + 11] = (256, 128, 1, 1, "bn")
        s[2 * i + 12] = (128, 256, 3, 1, "res")
    s[27] = (256, 512, 3, 2, "bn")                  # :222
    for i in range(8):                              # :226-236
        s[2 * i + 28] = (512, 256, 1, 1, "bn")
        s[2 * i + 29] = (256, 512, 3, 1, "res")
    s[44] = (512, 1024, 3, 2, "bn")                 # :240
    for i in range(4):                              # :244-254
        s[2 * i + 45] = (1024, 512, 1, 1, "bn")
        s[2 * i + 46] = (512, 1024, 3, 1, "res")
    for i in (53, 55, 57):                          # :258-272
        s[i] = (1024, 512, 1, 1, "bn")
    for i in (54, 56, 58):                          # :261-276
        s[i] = (512, 1024, 3, 1, "bn")
    s[59] = (1024, 24, 1, 1, "lin")                 # :277
    s[60] = (512, 256, 1, 1, "bn")                  # :285
    s[61] = (768, 256, 1, 1, "bn")                  # :293
    for i in (62, 64, 66):                          # :296-311
        s[i] = (256, 512, 3, 1, "bn")
    for i in (63, 65):
        s[i] = (512, 256, 1, 1, "bn")
    s[67] = (512, 24, 1, 1, "lin")                  # :312
    s[68] = (256, 128, 1, 1, "bn")                  # :320
    s[69] = (384, 128, 1, 1, "bn")                  # :328
    for i in (70, 72, 74):                          # :331-346
        s[i] = (128, 256, 3, 1, "bn")
    for i in (71, 73):
        s[i] = (256, 128, 1, 1, "bn")
    s[75] = (256, 24, 1, 1, "lin")                  # :347
    s[76] = (128, 64, 1, 1, "bn")                   # :381
    s[77] = (192, 64, 1, 1, "bn")                   # :389
    s[78] = (64, 128, 3, 1, "bn")                   # :392
    s[79] = (128, 32, 1, 1, "bn")                   # :396
    s[80] = (96, 32, 1, 1, "bn")                    # :404
    s[81] = (32, 64, 3, 1, "bn")                    # :407
    s[82] = (64, 9, 1, 1, "lin")                    # :410
    return s


def default_lock(stage: int = 1) -> Dict[int, bool]:
    """Stage 1 (shipped source): conv1-52 lock=True, 53-82 lock=False.
    Stage 2: everything unlocked (yolo/yolo3_net_pos.py:155-156, README.md:19)."""
    return {i: (stage == 1 and i <= 52) for i in range(1, 83)}


def _name(i: int, leaf: str) -> str:
    # variable names: train_yolo3_mask.py:86-103
    return "yolo/convolutional%d/%s" % (i, leaf)


def init_params(seed: int = 0, lock: Optional[Dict[int, bool]] = None,
                dtype=torch.float32, num_class: int = 3, k: int = K_MAP,
                xavier_locked: bool = False) -> Dict[str, torch.Tensor]:
    """Random-init variables with the reference's initialisers.

    Unlocked conv: xavier-uniform (yolo/yolo3_net_pos.py:118-119,138-139); locked
    conv: truncated_normal(0, 0.001) (:112,:135); gamma=1, beta=0, moving_mean=0,
    moving_variance=1 (:77-86); biases 0 (:115,:122).  ``xavier_locked`` draws the
    locked layers from xavier-uniform too: a stand-in for the pretrained backbone the
    reference restores over its 0.001-sigma placeholder init (train_yolo3_mask.py:75-107),
    without which 52 locked layers underflow to zero.
    """
    lock = lock if lock is not None else default_lock(1)
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, torch.Tensor] = {}
    for i, (cin, cout, ks, _, kind) in layer_specs().items():
        if kind == "lin":
            cout = (num_class + 5) * NUM_ANCHOR if i != 82 else k * k
        if lock[i] and not xavier_locked:
            w = torch.empty(ks, ks, cin, cout, dtype=torch.float64)
            torch.nn.init.trunc_normal_(w, 0.0, 0.001, -0.002, 0.002, generator=g)
        else:
            fan_in, fan_out = ks * ks * cin, ks * ks * cout
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w = (torch.rand(ks, ks, cin, cout, dtype=torch.float64, generator=g) * 2 - 1) * lim
        p[_name(i, "weights")] = w.to(dtype)
        if kind == "lin":
            p[_name(i, "biases")] = torch.zeros(cout, dtype=dtype)
        else:
            p[_name(i, "BatchNorm/gamma")] = torch.ones(cout, dtype=dtype)
            p[_name(i, "BatchNorm/beta")] = torch.zeros(cout, dtype=dtype)
            p[_name(i, "BatchNorm/moving_mean")] = torch.zeros(cout, dtype=dtype)
            p[_name(i, "BatchNorm/moving_variance")] = torch.ones(cout, dtype=dtype)
    return p


def trainable_names(lock: Dict[int, bool]) -> List[str]:
    """tf.trainable_variables() restated: unlocked weights, gamma, beta, biases
    (yolo/yolo3_net_pos.py:77-86,111-123,134-140)."""
    out = []
    for i, (_, _, _, _, kind) in layer_specs().items():
        if lock[i]:
            continue
        out.append(_name(i, "weights"))
        if kind == "lin":
            out.append(_name(i, "biases"))
        else:
            out.append(_name(i, "BatchNorm/gamma"))
            out.append(_name(i, "BatchNorm/beta"))
    return out


def regularized_names(lock: Dict[int, bool]) -> List[str]:
    """Variables carrying l2_regularizer(1e-4): unlocked conv weights and the
    biases of 59/67/75/82; never gamma/beta (yolo/yolo3_net_pos.py:38,120,123,140)."""
    return [n for n in trainable_names(lock) if n.endswith("weights") or n.endswith("biases")]


# ---------------------------------------------------------------------------
# primitives  (yolo/yolo3_net_pos.py:68-151)
# ---------------------------------------------------------------------------
def leaky_relu(x: torch.Tensor, alpha: float = ALPHA) -> torch.Tensor:
    """yolo/yolo3_net_pos.py:68-69  tf.maximum(alpha*x, x)."""
    return torch.maximum(alpha * x, x)


def same_pads(size: int, k: int, s: int) -> Tuple[int, int, int]:
    """[TF-sem] padding='SAME': out=ceil(size/s); total=max((out-1)*s+k-size,0);
    before=total//2, after=total-before (asymmetric for k=3,s=2, even size)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x: torch.Tensor, w_hwio: torch.Tensor, stride: int) -> torch.Tensor:
    """tf.nn.conv2d(NHWC, HWIO, strides=[1,s,s,1], padding='SAME')
    (yolo/yolo3_net_pos.py:125,142)."""
    k = w_hwio.shape[0]
    _, pt, pb = same_pads(x.shape[1], k, stride)
    _, pl, pr = same_pads(x.shape[2], k, stride)
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn, w_hwio.permute(3, 2, 0, 1), stride=stride)
    return y.permute(0, 2, 3, 1)


def bf16_ste(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (nearest-even) with a straight-through gradient.  NOT part of the
    reference (which is f32 throughout): passing ``quant=bf16_ste`` to build_network makes
    the oracle round weights and stored activations exactly where the bf16 HIP path does,
    so parity tests can use tolerances of a few bf16 ulps instead of the accumulated
    mixed-precision drift of 75 layers."""
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def fp8_e4m3(t: torch.Tensor, scale: float) -> torch.Tensor:
    """values representable as OCP e4m3 * scale: saturate to +-448, round to nearest even (torch's
    float8_e4m3fn conversion).  NOT part of the reference (f32 throughout): emulation of the port's fp8
    storage format for the parity tests of BASELINE.json configs[4]."""
    q = (t.detach().float() / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
    return q.to(t.dtype)


def _q(quant, t):
    return t if quant is None else quant(t)


def batch_norm(x: torch.Tensor, params: Dict[str, torch.Tensor], i: int, lock: bool,
               is_training: bool, updates: Optional[Dict[str, torch.Tensor]], quant=None) -> torch.Tensor:
    """yolo/yolo3_net_pos.py:71-107.

    lock: moving stats always (:76-81).  Otherwise training: batch moments over
    N,H,W with *population* variance (tf.nn.moments, :90 [TF-sem]); moving <-
    decay*moving + (1-decay)*batch (:93-96); normalise with batch stats (:98).
    Eval: moving stats (:101).  eps = 1e-5.
    """
    gamma = params[_name(i, "BatchNorm/gamma")]
    beta = params[_name(i, "BatchNorm/beta")]
    mm = params[_name(i, "BatchNorm/moving_mean")]
    mv = params[_name(i, "BatchNorm/moving_variance")]
    if lock or not is_training:
        mean, var = mm, mv
    else:
        mean = x.mean(dim=(0, 1, 2))
        var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
        if updates is not None:
            updates[_name(i, "BatchNorm/moving_mean")] = (mm * BN_DECAY + mean.detach() * (1 - BN_DECAY))
            updates[_name(i, "BatchNorm/moving_variance")] = (mv * BN_DECAY + var.detach() * (1 - BN_DECAY))
    # tf.nn.batch_normalization: (x-mean)*rsqrt(var+eps)*gamma + beta
    # (with ``quant`` the statistics come from the unrounded conv output and the
    #  normalisation reads the stored, rounded copy -- what the two-pass HIP path does)
    if quant is not None and not (lock or not is_training):
        x = quant(x)
    return (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta


def conv_bn(x, params, i, stride, lock, is_training, updates, alpha=ALPHA, quant=None):
    """yolo/yolo3_net_pos.py:132-146."""
    w = params[_name(i, "weights")]
    y = conv2d_same(x, w if i == 1 else _q(quant, w), stride)
    y = batch_norm(y, params, i, lock, is_training, updates, quant)
    return leaky_relu(y, alpha)


def conv_lin(x, params, i, quant=None):
    """yolo/yolo3_net_pos.py:109-130 with is_bias=True, is_act=False (all 4 call sites)."""
    return conv2d_same(x, _q(quant, params[_name(i, "weights")]), 1) + params[_name(i, "biases")]


def upsample2(x: torch.Tensor) -> torch.Tensor:
    """tf.image.resize_nearest_neighbor to 2x (yolo/yolo3_net_pos.py:290,325,386,401):
    out[y,x] = in[y>>1, x>>1] [TF-sem]."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def build_network(params: Dict[str, torch.Tensor], images: torch.Tensor, is_training: bool,
                  lock: Dict[int, bool], updates: Optional[Dict[str, torch.Tensor]] = None,
                  taps: Optional[Dict[str, torch.Tensor]] = None, quant=None,
                  force: Optional[Dict[str, torch.Tensor]] = None, fp8: Optional[Dict] = None):
    """yolo/yolo3_net_pos.py:153-463 (the active m=1/2 mask subnet, :380-412).

    Returns (yolos=[yolov3_3, yolov3_2, yolov3_1] each [B,g,g,3,5+C], mask_pos
    [B,S/2,S/2,k*k]).  ``taps`` (optional) collects intermediate activations by
    name ('act{i}') for per-layer parity tests.  ``force`` (optional, test aid) replaces
    the VALUE of the named activations after they are recorded in ``taps`` while keeping
    the autograd path (straight-through): each layer is then evaluated -- and later
    differentiated -- at the caller's linearisation point, so a per-layer comparison is not
    swamped by the chaotic amplification of bf16 rounding through 30 batch-stat BN layers
    of a randomly initialised network.
    """
    sp = layer_specs()
    # fp8 = {"from": 10, "upto": 52, "s_out": {layer: scale}, "s_w": {layer: scale}}: emulation of the port's e4m3 path --
    # layers from..upto (inference mode; "from" defaults to 1) store their output as e4m3 * s_out and use e4m3 * s_w
    # weights; layer from-1 (bf16) hands its bf16-rounded output over as e4m3 as well; the outputs the bf16 layers
    # consume too (skip2..5, act52) are ALSO kept as bf16 of the unquantised value
    dual = {}
    fp8_from = fp8.get("from", 1) if fp8 is not None else 1

    def cb(x, i):
        if fp8 is not None and max(2, fp8_from) <= i <= fp8["upto"]:
            p8 = dict(params)
            p8[_name(i, "weights")] = fp8_e4m3(params[_name(i, "weights")], fp8["s_w"][i])
            return conv_bn(x, p8, i, sp[i][3], lock[i], is_training, updates, quant=None)
        y = conv_bn(x, params, i, sp[i][3], lock[i], is_training, updates, quant=quant)
        return y

    def tap(name, t):
        # every stored activation is rounded once, after the residual add (bf16 path);
        # head logits / score maps stay f32
        idx = int(name[3:])
        if fp8 is not None and fp8_from <= idx <= fp8["upto"]:
            dual[name] = quant(t) if quant is not None else t
            t = fp8_e4m3(t, fp8["s_out"][idx])
        elif fp8 is not None and idx == fp8_from - 1:
            # the bf16 layer in front of the fp8 range: stored in bf16 (what its bf16 consumers read), quantised FROM that
            dual[name] = quant(t) if quant is not None else t
            t = fp8_e4m3(dual[name], fp8["s_out"][idx])
        elif quant is not None and name not in ("act59", "act67", "act75", "act82"):
            t = quant(t)
        if taps is not None:
            taps[name] = t
        if force is not None and name in force:
            t = t + (force[name].to(t.dtype) - t).detach()
        return t

    net = tap("act1", cb(images, 1))                                     # :159
    net = tap("act2", cb(net, 2))                                        # :165
    skips = {}
    i = 3
    for nblocks, skipname, down in ((1, "skip2", 5), (2, "skip3", 10), (8, "skip4", 27),
                                    (8, "skip5", 44), (4, None, None)):
        for _ in range(nblocks):
            shortcut = net
            net = tap("act%d" % i, cb(net, i))                           # 1x1
            net = tap("act%d" % (i + 1), cb(net, i + 1) + shortcut)      # res add after act (:148-151)
            i += 2
        if skipname:
            skips[skipname] = dual.get("act%d" % (i - 1), net)
            net = tap("act%d" % down, cb(net, down))
            i = down + 1
    net = dual.get("act52", net)
    # head 1 (:258-281)
    for i in (53, 54, 55, 56, 57):
        net = tap("act%d" % i, cb(net, i))
    y1 = tap("act58", cb(net, 58))
    y1 = tap("act59", conv_lin(y1, params, 59, quant))
    B = images.shape[0]
    yolov3_1 = y1.reshape(B, y1.shape[1], y1.shape[2], NUM_ANCHOR, -1)
    # head 2 (:285-316)
    net = tap("act60", cb(net, 60))
    net = torch.cat([skips["skip5"], upsample2(net)], dim=-1)            # :290-291 [skip, up]
    for i in (61, 62, 63, 64, 65):
        net = tap("act%d" % i, cb(net, i))
    y2 = tap("act66", cb(net, 66))
    y2 = tap("act67", conv_lin(y2, params, 67, quant))
    yolov3_2 = y2.reshape(B, y2.shape[1], y2.shape[2], NUM_ANCHOR, -1)
    # head 3 (:320-351)
    net = tap("act68", cb(net, 68))
    net = torch.cat([skips["skip4"], upsample2(net)], dim=-1)            # :325-326
    for i in (69, 70, 71, 72, 73):
        net = tap("act%d" % i, cb(net, i))
    y3 = tap("act74", cb(net, 74))
    y3 = tap("act75", conv_lin(y3, params, 75, quant))
    yolov3_3 = y3.reshape(B, y3.shape[1], y3.shape[2], NUM_ANCHOR, -1)
    # mask subnet m=1/2 (:381-412), branches from conv73's output
    net = tap("act76", cb(net, 76))
    net = torch.cat([skips["skip3"], upsample2(net)], dim=-1)            # :386-387
    net = tap("act77", cb(net, 77))
    net = tap("act78", cb(net, 78))
    net = tap("act79", cb(net, 79))
    net = torch.cat([skips["skip2"], upsample2(net)], dim=-1)            # :401-402
    net = tap("act80", cb(net, 80))
    net = tap("act81", cb(net, 81))
    mask_pos = tap("act82", conv_lin(net, params, 82, quant))                   # :410
    return [yolov3_3, yolov3_2, yolov3_1], mask_pos


# ---------------------------------------------------------------------------
# decode  (yolo/yolo3_net_pos.py:465-514)
# ---------------------------------------------------------------------------
def cell_offset(gh: int, gw: int, dtype) -> torch.Tensor:
    """self.offset[:, :gh, :gw] (yolo/yolo3_net_pos.py:23-26): [...,0]=x index, [...,1]=y index."""
    cx = torch.arange(gw, dtype=dtype).view(1, 1, gw, 1, 1).expand(1, gh, gw, NUM_ANCHOR, 1)
    cy = torch.arange(gh, dtype=dtype).view(1, gh, 1, 1, 1).expand(1, gh, gw, NUM_ANCHOR, 1)
    return torch.cat([cx, cy], dim=-1)


def interpret_output(predicts: Sequence[torch.Tensor], anchors=ANCHORS):
    """yolo/yolo3_net_pos.py:465-514.  predicts = [72-grid, 36-grid, 18-grid]."""
    dtype = predicts[0].dtype
    net_h = predicts[2].shape[1] * 32
    net_w = predicts[2].shape[2] * 32
    net_factor = torch.tensor([net_w, net_h], dtype=dtype).view(1, 1, 1, 1, 2)
    anchors_pwhs, conf_logits, class_logits, pred_coords, pred_norm_coords = [], [], [], [], []
    for i in (0, 1, 2):
        preds = predicts[i]
        gh, gw = preds.shape[1], preds.shape[2]
        grid_factor = torch.tensor([gw, gh], dtype=dtype).view(1, 1, 1, 1, 2)
        pred_conf = preds[..., 4:5]
        pred_class = preds[..., 5:]
        pred_cxy = torch.sigmoid(preds[..., :2])
        pred_twh = preds[..., 2:4]
        box_xy = cell_offset(gh, gw, dtype) + pred_cxy
        a = torch.tensor(np.asarray(anchors, dtype=np.float64)[3 * i:3 * i + 3], dtype=dtype)  # [3,2] (w,h)
        anchors_pwh = a.view(1, 1, 1, NUM_ANCHOR, 2).expand(preds.shape[0], gh, gw, NUM_ANCHOR, 2)
        box_wh = torch.exp(pred_twh) * anchors_pwh
        pred_norm_coord = torch.cat([box_xy / grid_factor, box_wh / net_factor], dim=-1)
        anchors_pwhs.append(anchors_pwh)
        conf_logits.append(pred_conf)
        class_logits.append(pred_class)
        pred_coords.append(torch.cat([pred_cxy, pred_twh], dim=-1))
        pred_norm_coords.append(pred_norm_coord)
    return [[net_h, net_w], anchors_pwhs, conf_logits, class_logits, pred_coords, pred_norm_coords]


# ---------------------------------------------------------------------------
# detection filter  (yolo/yolo3_net_pos.py:517-628, 940-952)
# ---------------------------------------------------------------------------
def clip_boxes(boxes: np.ndarray, window: np.ndarray) -> np.ndarray:
    """clip_boxes_graph, yolo/yolo3_net_pos.py:940-952: max(min(v, hi), lo)."""
    wy1, wx1, wy2, wx2 = window
    out = boxes.copy()
    out[:, 0] = np.maximum(np.minimum(boxes[:, 0], wy2), wy1)
    out[:, 1] = np.maximum(np.minimum(boxes[:, 1], wx2), wx1)
    out[:, 2] = np.maximum(np.minimum(boxes[:, 2], wy2), wy1)
    out[:, 3] = np.maximum(np.minimum(boxes[:, 3], wx2), wx1)
    return out


def _tf_iou(a: np.ndarray, b: np.ndarray) -> float:
    """[TF-sem] IoU as tf.image.non_max_suppression computes it (f32): corner order
    normalised with min/max; zero-area box => IoU 0."""
    f = np.float32
    ya1, xa1, ya2, xa2 = f(min(a[0], a[2])), f(min(a[1], a[3])), f(max(a[0], a[2])), f(max(a[1], a[3]))
    yb1, xb1, yb2, xb2 = f(min(b[0], b[2])), f(min(b[1], b[3])), f(max(b[0], b[2])), f(max(b[1], b[3]))
    area_a = f(f(ya2 - ya1) * f(xa2 - xa1))
    area_b = f(f(yb2 - yb1) * f(xb2 - xb1))
    if area_a <= 0 or area_b <= 0:
        return 0.0
    iy1, ix1, iy2, ix2 = max(ya1, yb1), max(xa1, xb1), min(ya2, yb2), min(xa2, xb2)
    inter = f(max(f(iy2 - iy1), f(0)) * max(f(ix2 - ix1), f(0)))
    return float(f(inter / f(f(area_a + area_b) - inter)))


def non_max_suppression(boxes: np.ndarray, scores: np.ndarray, max_out: int, iou_thresh: float) -> List[int]:
    """[TF-sem] tf.image.non_max_suppression: visit candidates by descending score
    (ties: lower index first -- not verifiable without TF); keep a box
    unless its IoU with an already kept box is > iou_thresh; stop at max_out."""
    order = sorted(range(len(scores)), key=lambda j: (-float(scores[j]), j))
    keep: List[int] = []
    for j in order:
        if len(keep) >= max_out:
            break
        if all(_tf_iou(boxes[j], boxes[q]) <= iou_thresh for q in keep):
            keep.append(j)
    return keep


def filter_detections(conf_logit, class_logit, pred_norm_coord, batch_window, obj_thresh=OBJ_THRESHOLD,
                      nms_thresh=IOU_THRESHOLD, max_detection=MAX_DETECTION, return_index: bool = False):
    """yolo/yolo3_net_pos.py:517-628.  Returns float32 [B, max_detection, 6]
    rows (y1,x1,y2,x2,classid,score), score-descending, zero padded.  All scoring
    arithmetic is done in f32 like the reference graph.  ``return_index`` (test aid, not in the
    reference): also the candidate index of every row (position in the 72,36,18-grid concatenation
    of :527-538, each grid flattened (y, x, anchor)), -1 where padded."""
    B = conf_logit[0].shape[0]
    out = np.zeros((B, max_detection, 6), dtype=np.float32)
    out_idx = np.full((B, max_detection), -1, dtype=np.int64)
    for i in range(B):
        confs, clss, boxes = [], [], []
        for j in (0, 1, 2):                                             # :527-538 order 72,36,18
            confs.append(torch.sigmoid(conf_logit[j][i].detach().float()).reshape(-1))
            clss.append(torch.softmax(class_logit[j][i].detach().float(), dim=-1).reshape(-1, class_logit[j].shape[-1]))
            boxes.append(pred_norm_coord[j][i].detach().float().reshape(-1, 4))
        pred_conf = torch.cat(confs).numpy()
        pred_class = torch.cat(clss).numpy()
        box = torch.cat(boxes).numpy()
        classid = np.argmax(pred_class, axis=-1).astype(np.int32)       # :545 (first max wins)
        classmax = pred_class[np.arange(pred_class.shape[0]), classid]
        score = (pred_conf * classmax).astype(np.float32)               # :548
        xc, yc, w, h = box[:, 0], box[:, 1], box[:, 2], box[:, 3]
        half = np.float32(2.0)
        yxyx = np.stack([yc - h / half, xc - w / half, yc + h / half, xc + w / half], axis=-1).astype(np.float32)
        yxyx = clip_boxes(yxyx, np.asarray(batch_window[i], dtype=np.float32))      # :552-555
        keep = np.where(score > np.float32(obj_thresh))[0]              # :558
        nms_keep: List[int] = []
        for c in np.unique(classid[keep]):                              # :565-589 per-class NMS
            ixs = keep[classid[keep] == c]
            ck = non_max_suppression(yxyx[ixs], score[ixs], max_detection, nms_thresh)
            nms_keep.extend(int(ixs[q]) for q in ck)
        keep = np.array(sorted(set(nms_keep)), dtype=np.int64)          # :590-592 set_intersection => ascending ids
        num_keep = min(len(keep), max_detection)                        # :608-612 top_k, ties: lower index first
        order = sorted(range(len(keep)), key=lambda q: (-float(score[keep[q]]), q))[:num_keep]
        keep = keep[order] if len(keep) else keep
        for r, idx in enumerate(keep):
            out[i, r, :4] = yxyx[idx]
            out[i, r, 4] = np.float32(classid[idx])
            out[i, r, 5] = score[idx]
            out_idx[i, r] = idx
    return (out, out_idx) if return_index else out


# ---------------------------------------------------------------------------
# YOLO loss  (yolo/yolo3_net_pos.py:631-747)
# ---------------------------------------------------------------------------
def sigmoid_ce(labels: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """[TF-sem] tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|)), written the
    way TF builds it -- relu = where(x >= 0, x, 0), -|x| = where(x >= 0, -x, x) -- so that autograd at
    x == 0 exactly gives TF's gradient sigmoid(0) - z = 0.5 - z.  (max/abs would give the subgradient
    1 - z there: |x|' = 0 in torch.  Found by the GPU known-answer test with logits that are exactly 0.)"""
    cond = logits >= 0
    relu = torch.where(cond, logits, torch.zeros_like(logits))
    neg_abs = torch.where(cond, -logits, logits)
    return relu - logits * labels + torch.log1p(torch.exp(neg_abs))


def loss_yolo(predicts, true_boxes: torch.Tensor, labels_value: Sequence[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """yolo/yolo3_net_pos.py:631-747.  labels_value = [yolo3, yolo2, yolo1] (:55)."""
    (net_h, net_w), anchors_pwhs, conf_logits, class_logits, pred_coords, pred_norm_coords = predicts
    dtype = conf_logits[0].dtype
    acc = {k: torch.zeros((), dtype=dtype) for k in ("obj", "noobj", "xy", "wh", "conf", "class", "coord")}
    for i in range(3):
        pnc = pred_norm_coords[i]
        pred_xy = pnc[..., :2].unsqueeze(4)
        pred_wh = pnc[..., 2:4].unsqueeze(4)
        pred_mins, pred_maxes = pred_xy - pred_wh / 2.0, pred_xy + pred_wh / 2.0
        true_xy, true_wh = true_boxes[..., 0:2], true_boxes[..., 2:4]
        true_mins, true_maxes = true_xy - true_wh / 2.0, true_xy + true_wh / 2.0
        iwh = torch.clamp(torch.minimum(pred_maxes, true_maxes) - torch.maximum(pred_mins, true_mins), min=0.0)
        inter = iwh[..., 0] * iwh[..., 1]
        true_areas = true_wh[..., 0] * true_wh[..., 1]
        pred_areas = pred_wh[..., 0] * pred_wh[..., 1]
        union = torch.clamp(pred_areas + true_areas - inter, min=1e-10)            # :676
        iou = torch.clamp(inter / union, 0.0, 1.0)
        best = iou.max(dim=4).values
        ignore = (best < IGNORE_THRESH).to(dtype).unsqueeze(4)                     # :680
        obj_val = labels_value[i]
        pred_conf = conf_logits[i]
        obj_mask = obj_val[..., 4:5]
        noobj_mask = 1.0 - obj_mask
        ce = sigmoid_ce(obj_mask, pred_conf)
        object_loss = (obj_mask * ce * OBJECT_SCALE).sum(dim=(1, 2, 3, 4)).mean()                  # :691-692
        noobject_loss = (ignore * noobj_mask * ce * NOOBJECT_SCALE).sum(dim=(1, 2, 3, 4)).mean()   # :693-694
        true_cls = obj_val[..., 5:].argmax(dim=-1)                                                 # :699
        logp = torch.log_softmax(class_logits[i], dim=-1)
        sce = -logp.gather(-1, true_cls.unsqueeze(-1))                                             # :701
        class_loss = (obj_mask * sce * CLASS_SCALE).sum(dim=(1, 2, 3, 4)).mean()
        gh, gw = pred_conf.shape[1], pred_conf.shape[2]
        grid_factor = torch.tensor([gw, gh], dtype=dtype).view(1, 1, 1, 1, 2)
        net_factor = torch.tensor([net_w, net_h], dtype=dtype).view(1, 1, 1, 1, 2)
        tb = obj_val[..., 0:4]
        true_cxy = tb[..., 0:2] * grid_factor - cell_offset(gh, gw, dtype)                         # :714
        true_twh = torch.clamp(torch.log(tb[..., 2:4] * net_factor / anchors_pwhs[i]), -1e2, 1e2)  # :717-718
        wh_scale = (2.0 - tb[..., 2] * tb[..., 3]).unsqueeze(4)                                    # :721-722
        cxy_delta = obj_mask * (pred_coords[i][..., :2] - true_cxy)
        twh_delta = obj_mask * (pred_coords[i][..., 2:4] - true_twh)
        xy_loss = (cxy_delta ** 2 * wh_scale ** 2 * COORD_SCALE).sum(dim=(1, 2, 3, 4)).mean()      # :725
        wh_loss = (twh_delta ** 2 * wh_scale ** 2 * COORD_SCALE).sum(dim=(1, 2, 3, 4)).mean()      # :726
        acc["obj"] = acc["obj"] + object_loss
        acc["noobj"] = acc["noobj"] + noobject_loss
        acc["xy"] = acc["xy"] + xy_loss
        acc["wh"] = acc["wh"] + wh_loss
        acc["conf"] = acc["conf"] + object_loss + noobject_loss
        acc["class"] = acc["class"] + class_loss
        acc["coord"] = acc["coord"] + xy_loss + wh_loss
    return acc


# ---------------------------------------------------------------------------
# position-sensitive assembly + mask loss  (yolo/yolo3_net_pos.py:750-860, 954-975)
# ---------------------------------------------------------------------------
def overlaps(boxes1: np.ndarray, boxes2: np.ndarray) -> np.ndarray:
    """overlaps_graph, yolo/yolo3_net_pos.py:954-975 (f32, no epsilon on the union)."""
    b1 = boxes1.astype(np.float32)[:, None, :]
    b2 = boxes2.astype(np.float32)[None, :, :]
    y1 = np.maximum(b1[..., 0], b2[..., 0]); x1 = np.maximum(b1[..., 1], b2[..., 1])
    y2 = np.minimum(b1[..., 2], b2[..., 2]); x2 = np.minimum(b1[..., 3], b2[..., 3])
    inter = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0)
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / (a1 + a2 - inter)).astype(np.float32)


def kmask_edges(lo: float, hi: float, k: int = K_MAP) -> List[int]:
    """Bin edges along one axis, yolo/yolo3_net_pos.py:804-813: [int(lo),
    round(lo + j*(hi-lo)/k) for j=1..k-1, int(hi)], f32 arithmetic, tf.round =
    half-to-even, tf.cast(int32) = truncation [TF-sem]."""
    f = np.float32
    lo, hi = f(lo), f(hi)
    sub = f(f(hi - lo) / f(k))
    edges = [int(lo)]
    for j in range(1, k):
        edges.append(int(np.round(f(lo + f(f(j) * sub)))))
    edges.append(int(hi))
    return edges


def channel_index_map(box: Sequence[float], size: int, k: int = K_MAP) -> np.ndarray:
    """assemble_kmask_from_box, yolo/yolo3_net_pos.py:799-839, as an index map:
    int32 [size,size], value = by*k+bx inside bin (by,bx), -1 elsewhere."""
    gy = kmask_edges(box[0], box[2], k)
    gx = kmask_edges(box[1], box[3], k)
    m = -np.ones((size, size), dtype=np.int32)
    for by in range(k):
        for bx in range(k):
            y1, y2, x1, x2 = gy[by], gy[by + 1], gx[bx], gx[bx + 1]
            if y2 > y1 and x2 > x1:
                m[max(y1, 0):min(y2, size), max(x1, 0):min(x2, size)] = by * k + bx
    return m


def assemble_logits(score_maps: torch.Tensor, box: Sequence[float], k: int = K_MAP):
    """reduce_sum(pred_masks * channel_masks, -1) for one box
    (yolo/yolo3_net_pos.py:843-848): returns (logits [S,S] -- 0 outside the box,
    mask_object [S,S] in {0,1})."""
    size = score_maps.shape[0]
    idx = torch.from_numpy(channel_index_map(box, size, k)).long()
    inside = idx >= 0
    logits = torch.gather(score_maps, 2, idx.clamp(min=0).unsqueeze(-1)).squeeze(-1)
    logits = torch.where(inside, logits, torch.zeros_like(logits))
    return logits, inside.to(score_maps.dtype)


def select_mask_rois(detections_i: np.ndarray, true_boxes_i: np.ndarray,
                     perm_det: Optional[Sequence[int]] = None, perm_gt: Optional[Sequence[int]] = None):
    """yolo/yolo3_net_pos.py:757-796.  Returns (positive_rois [P,4] normalised yxyx,
    gt_assignment [P] index into the *trimmed* GT list, gt_index_map: trimmed->row).

    tf.random_shuffle (:781-782) is replaced by the injected permutations
    (identity when None) -- the reference is non-deterministic here.
    """
    prop = detections_i[:, :4].astype(np.float32)
    prop = prop[np.abs(prop).sum(axis=1) != 0]                                         # :759-760
    gt = true_boxes_i[:, :4].astype(np.float32)
    gt_rows = np.where(np.abs(gt).sum(axis=1) != 0)[0]                                 # :766-769
    gt = gt[gt_rows]
    xc, yc, w, h = gt[:, 0], gt[:, 1], gt[:, 2], gt[:, 3]
    two = np.float32(2.0)
    gt_yxyx = np.stack([yc - h / two, xc - w / two, yc + h / two, xc + w / two], axis=-1).astype(np.float32)  # :778-779
    pd = list(range(len(prop))) if perm_det is None else [q for q in perm_det if q < len(prop)]
    pg = list(range(len(gt_yxyx))) if perm_gt is None else [q for q in perm_gt if q < len(gt_yxyx)]
    rois = np.concatenate([prop[pd][:7], gt_yxyx[pg][:3]], axis=0)                     # :783
    if len(rois) == 0 or len(gt_yxyx) == 0:
        return np.zeros((0, 4), np.float32), np.zeros((0,), np.int64), gt_rows
    ov = overlaps(rois, gt_yxyx)                                                       # :784
    iou_max = ov.max(axis=1)
    pos = np.where(iou_max >= np.float32(0.5))[0]                                      # :787-789
    return rois[pos], ov[pos].argmax(axis=1), gt_rows


def loss_mask(detections: np.ndarray, mask_pos: torch.Tensor, true_boxes: np.ndarray, true_masks: np.ndarray,
              perms: Optional[Sequence[Tuple[Sequence[int], Sequence[int]]]] = None) -> torch.Tensor:
    """yolo/yolo3_net_pos.py:750-860.  detections [B,30,6] (constant wrt autograd, the
    boxes pass through tf.round), mask_pos [B,S,S,k*k] torch, true_boxes
    [B,1,1,1,20,5], true_masks bool [B,20,2S,2S]."""
    B, size = mask_pos.shape[0], mask_pos.shape[1]
    dtype = mask_pos.dtype
    total = torch.zeros((), dtype=dtype)
    for i in range(B):
        pd, pg = perms[i] if perms is not None else (None, None)
        pos_rois, assign, gt_rows = select_mask_rois(detections[i], true_boxes[i, 0, 0, 0], pd, pg)
        if len(pos_rois) == 0:                                                         # :855
            continue
        # GT masks -> score-map size: legacy bilinear at exact 2x == [::2, ::2], then round (:771-775) [TF-sem]
        step = true_masks.shape[2] // size
        gt_small = true_masks[i][gt_rows][:, ::step, ::step].astype(np.float32)
        rois_px = np.round(pos_rois * np.float32(size))                                # :842 half-to-even
        per_roi = []
        for r in range(len(rois_px)):
            logits, mobj = assemble_logits(mask_pos[i], rois_px[r])
            gtm = torch.from_numpy(gt_small[assign[r]]).to(dtype)
            num = (mobj * sigmoid_ce(gtm, logits)).sum()                               # :850
            per_roi.append(num / mobj.sum())                                           # :852 (0/0 -> nan)
        total = total + MASK_SCALE * torch.stack(per_roi).mean()
    return total / B                                                                   # :858


def val_test(detections: np.ndarray, mask_pos: torch.Tensor):
    """yolo/yolo3_net_pos.py:862-938.  Returns (det_box list of [n,6] f32, det_mask
    list of [n,S,S] f32 or scalar 0.0).  Outside the box the value is sigmoid(0)=0.5."""
    det_box, det_mask = [], []
    B, size = mask_pos.shape[0], mask_pos.shape[1]
    for i in range(B):
        prop = detections[i].astype(np.float32)
        pb = np.round(prop[:, :4] * np.float32(size))                                  # :876
        keep = np.where(((pb[:, 2] - pb[:, 0]) > 0) & ((pb[:, 3] - pb[:, 1]) > 0))[0]  # :877-878
        prop, pb = prop[keep], pb[keep]
        if prop.size > 0:
            masks = [torch.sigmoid(assemble_logits(mask_pos[i], pb[r])[0]) for r in range(len(pb))]
            det_mask.append(torch.stack(masks).float().numpy())
        else:
            det_mask.append(np.float32(0.0))                                           # :933
        det_box.append(prop)
    return det_box, det_mask


# ---------------------------------------------------------------------------
# total loss + optimizer
# ---------------------------------------------------------------------------
def l2_regularization(params: Dict[str, torch.Tensor], lock: Dict[int, bool]) -> torch.Tensor:
    """[TF-sem] tf.contrib.layers.l2_regularizer(1e-4)(w) = 1e-4 * sum(w^2)/2, added by
    tf.losses.get_total_loss() (yolo/yolo3_net_pos.py:38,61)."""
    tot = None
    for n in regularized_names(lock):
        t = L2_WEIGHT * 0.5 * (params[n] ** 2).sum()
        tot = t if tot is None else tot + t
    return tot


def total_loss(params, batch, lock, is_training=True, perms=None, updates=None, obj_thresh=OBJ_THRESHOLD,
               quant=None, taps=None, force=None):
    """YOLONet.__init__ wiring, yolo/yolo3_net_pos.py:47-61.  ``batch`` keys: images,
    clip_window, true_boxes, true_masks, yolo1, yolo2, yolo3 (torch / numpy)."""
    yolos, mask_pos = build_network(params, batch["images"], is_training, lock, updates, taps, quant, force)
    pred = interpret_output(yolos)
    with torch.no_grad():
        det = filter_detections(pred[2], pred[3], pred[5], batch["clip_window"], obj_thresh)
    ly = loss_yolo(pred, batch["true_boxes"], [batch["yolo3"], batch["yolo2"], batch["yolo1"]])
    lm = loss_mask(det, mask_pos, batch["true_boxes"].detach().numpy(), batch["true_masks"], perms)
    reg = l2_regularization(params, lock)
    parts = dict(ly)
    parts["mask"] = lm
    parts["reg"] = reg
    parts["total"] = ly["conf"] + ly["class"] + ly["coord"] + lm + reg
    return parts, det, yolos, mask_pos


def adam_tf_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, t: int,
                 lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """[TF-sem] tf.train.AdamOptimizer (train_yolo3_mask.py:55): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2; p <- p - lr_t * m / (sqrt(v) + eps).  t starts at 1."""
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    p = p - lr_t * m / (torch.sqrt(v) + eps)
    return p, m, v


# ---------------------------------------------------------------------------
# synthetic batch incl. the target assignment of utils/train_data.py:149-178
# ---------------------------------------------------------------------------
def assign_targets(boxes_px: np.ndarray, cls: np.ndarray, S: int, anchors=ANCHORS, num_class=3):
    """utils/train_data.py:149-178: best-IoU anchor of the 9 (centred boxes), cell =
    int(xc*g/net), skip if occupied.  boxes_px rows (xc,yc,w,h) in pixels.
    Returns yolo3 (S/8 grid), yolo2, yolo1 with pixel coords (caller divides by S, :258-261)."""
    g1 = S // 32
    yolos = [np.zeros((4 * g1, 4 * g1, 3, 5 + num_class), np.float32),
             np.zeros((2 * g1, 2 * g1, 3, 5 + num_class), np.float32),
             np.zeros((g1, g1, 3, 5 + num_class), np.float32)]
    amax = np.asarray(anchors, np.float32) / 2.0
    a_area = amax[:, 0] * amax[:, 1] * 4
    for b, c in zip(boxes_px, cls):
        half = np.asarray(b[2:4], np.float32) / 2.0
        imax = np.minimum(half[None, :], amax)
        inter = np.maximum(2 * imax, 0.0)
        ia = inter[:, 0] * inter[:, 1]
        iou = ia / (half[0] * half[1] * 4 + a_area - ia)
        if iou.max() <= 0:
            continue
        idx = int(np.argmax(iou))
        y = yolos[idx // 3]
        xi = int(b[0] * y.shape[1] / S)
        yi = int(b[1] * y.shape[0] / S)
        if y[yi, xi, idx % 3, 4] == 1:
            continue
        y[yi, xi, idx % 3, 0:4] = b[:4]
        y[yi, xi, idx % 3, 4] = 1
        y[yi, xi, idx % 3, 5 + int(c)] = 1.0
    return yolos


def synthetic_batch(B: int, S: int, seed: int = 1234, dtype=torch.float32, num_class: int = 3):
    """Seeded synthetic training batch: uniform images, 1..5 ellipse
    instances per image with tight boxes, masks [B,20,S,S], targets via assign_targets."""
    rng = np.random.RandomState(seed)
    images = rng.rand(B, S, S, 3).astype(np.float32)
    true_boxes = np.zeros((B, 1, 1, 1, MAX_BOX_PER_IMAGE, 5), np.float32)
    true_masks = np.zeros((B, MAX_BOX_PER_IMAGE, S, S), bool)
    y3 = np.zeros((B, S // 8, S // 8, 3, 5 + num_class), np.float32)
    y2 = np.zeros((B, S // 16, S // 16, 3, 5 + num_class), np.float32)
    y1 = np.zeros((B, S // 32, S // 32, 3, 5 + num_class), np.float32)
    yy, xx = np.mgrid[0:S, 0:S]
    for b in range(B):
        n = rng.randint(1, 6)
        bx, cl = [], []
        for j in range(n):
            w = rng.uniform(0.05, 0.6) * S
            h = rng.uniform(0.05, 0.6) * S
            cx = rng.uniform(w / 2, S - w / 2)
            cy = rng.uniform(h / 2, S - h / 2)
            m = ((xx - cx) / (w / 2)) ** 2 + ((yy - cy) / (h / 2)) ** 2 <= 1.0
            if not m.any():
                continue
            ys, xs = np.where(m)
            x1, x2, y1_, y2_ = xs.min(), xs.max(), ys.min(), ys.max()
            true_masks[b, j] = m
            c = rng.randint(0, num_class)
            box = [(x1 + x2) / 2.0, (y1_ + y2_) / 2.0, float(x2 - x1), float(y2_ - y1_)]
            if box[2] <= 0 or box[3] <= 0:
                true_masks[b, j] = False
                continue
            true_boxes[b, 0, 0, 0, j, :4] = np.asarray(box, np.float32) / S
            true_boxes[b, 0, 0, 0, j, 4] = c
            bx.append(box)
            cl.append(c)
        t3, t2, t1 = assign_targets(np.asarray(bx, np.float32).reshape(-1, 4), np.asarray(cl), S, num_class=num_class)
        for t in (t3, t2, t1):
            t[..., 0:4] /= S
        y3[b], y2[b], y1[b] = t3, t2, t1
    window = np.tile(np.array([[0.0, 0.0, 1.0, 1.0]], np.float32), (B, 1))
    return {
        "images": torch.from_numpy(images).to(dtype),
        "clip_window": window,
        "true_boxes": torch.from_numpy(true_boxes).to(dtype),
        "true_masks": true_masks,
        "yolo1": torch.from_numpy(y1).to(dtype),
        "yolo2": torch.from_numpy(y2).to(dtype),
        "yolo3": torch.from_numpy(y3).to(dtype),
    }


# ---------------------------------------------------------------------------------------------
# evaluate(): host post-processing of sess.run(net.evaluation) -- SURVEY.md 8(f3)
# ---------------------------------------------------------------------------------------------
def correct_yolo_boxes(x1, y1, x2, y2, image_h: int, image_w: int, net_h: int, net_w: int):
    """calculate_test_map.py:121-138: undo the letter box, normalised box -> integer pixel corners
    of the original image (np.around = half-to-even, clamped to the image).  Pinned by
    tests/golden/correct_yolo_boxes.json (the reference's own function executed on the four
    sample image sizes)."""
    if (float(net_w) / image_w) < (float(net_h) / image_h):
        new_w = net_w
        new_h = (image_h * net_w) // image_w
    else:
        new_h = net_h
        new_w = (image_w * net_h) // image_h
    x_offset, x_scale = float((net_w - new_w) // 2) / net_w, float(new_w) / net_w
    y_offset, y_scale = float((net_h - new_h) // 2) / net_h, float(new_h) / net_h

    def corner(v, off, scale, n):
        return max(min(int(np.around((v - off) / scale * n).astype(np.int32)), n), 0)
    return (corner(np.float32(x1), x_offset, x_scale, image_w), corner(np.float32(y1), y_offset, y_scale, image_h),
            corner(np.float32(x2), x_offset, x_scale, image_w), corner(np.float32(y2), y_offset, y_scale, image_h))


def letterbox_window(image_h: int, image_w: int, size: int) -> np.ndarray:
    """the clip window of image_read (calculate_test_map.py:151-169): [top, left, bottom, right] of the
    resized image inside the size x size letter box, normalised"""
    imgh, imgw = image_h, image_w
    if (float(size) / imgw) < (float(size) / imgh):
        imgh = (imgh * size) // imgw
        imgw = size
    else:
        imgw = (imgw * size) // imgh
        imgh = size
    top, left = (size - imgh) // 2, (size - imgw) // 2
    return np.array([top / size, left / size, (imgh + top) / size, (imgw + left) / size], np.float32)


def resize_linear(src: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """cv2.resize(src, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR) for a float32 2-D array, restated
    from OpenCV's documented behaviour (cv2 itself is not installed here: unpinned): pixel centres
    aligned (src = (dst + 0.5) * scale - 0.5), the source index clamped at both borders with weight
    0 on the out-of-range neighbour, horizontal pass then vertical pass, all in float32."""
    src = np.asarray(src, np.float32)
    sh, sw = src.shape

    def taps(dn, sn):
        scale = np.float64(sn) / dn
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        a = (f - s0.astype(np.float32)).astype(np.float32)
        lo = s0 < 0
        a[lo], s0[lo] = 0.0, 0
        hi = s0 >= sn - 1
        a[hi], s0[hi] = 0.0, sn - 1
        return s0, np.minimum(s0 + 1, sn - 1), a
    x0, x1, ax = taps(dst_w, sw)
    y0, y1, ay = taps(dst_h, sh)
    one = np.float32(1.0)
    rows = src[:, x0] * (one - ax)[None, :] + src[:, x1] * ax[None, :]
    return (rows[y0, :] * (one - ay)[:, None] + rows[y1, :] * ay[:, None]).astype(np.float32)


def image_read(image_rgb: np.ndarray, image_size: int):
    """calculate_test_map.py:149-176 (= utils/val_data.py:36-63): letter box an RGB uint8 image: resize
    with cv2.INTER_LINEAR on the float32 image (restated by resize_linear, per channel), centre it on
    a 127 canvas, divide by 255.0 (float64, as numpy does), return (image f32 [S,S,3], window)."""
    imgh, imgw = image_rgb.shape[:2]
    if (float(image_size) / imgw) < (float(image_size) / imgh):
        imgh = (imgh * image_size) // imgw
        imgw = image_size
    else:
        imgw = (imgw * image_size) // imgh
        imgh = image_size
    img = image_rgb.astype(np.float32)
    small = np.stack([resize_linear(img[:, :, c], imgw, imgh) for c in range(3)], axis=-1)
    canvas = np.ones((image_size, image_size, 3)) * 127.0
    top, left = (image_size - imgh) // 2, (image_size - imgw) // 2
    canvas[top:top + imgh, left:left + imgw, :] = small
    return (canvas / 255.0).astype(np.float32), letterbox_window(image_rgb.shape[0], image_rgb.shape[1], image_size)


def paste_detections(det_box: np.ndarray, det_mask, image_h: int, image_w: int, net_size: int):
    """the per-image body of evaluate (calculate_test_map.py:220-269): every detection's mask is cut
    out of the size x size map at its rounded box, resized to the un-letterboxed box, thresholded at
    0.5 and pasted into an image-sized boolean mask; `merged` gets classid+1 where a mask is set,
    later detections overwriting earlier ones.  Returns (entries, merged): entries = list of dicts
    {index, classid, score, mask}.  A detection whose crop is empty is skipped (the reference would
    raise inside cv2.resize there)."""
    merged = np.zeros((image_h, image_w), np.uint8)
    entries = []
    if np.isscalar(det_mask) or np.sum(det_mask) == 0.0:
        return entries, merged
    for k in range(det_box.shape[0]):
        y1n, x1n, y2n, x2n = (np.float32(v) for v in det_box[k, :4])
        x1, y1, x2, y2 = correct_yolo_boxes(x1n, y1n, x2n, y2n, image_h, image_w, net_size, net_size)
        if (y2 - y1) * (x2 - x1) <= 0:
            continue
        size = det_mask[k].shape[0]
        cy1, cx1, cy2, cx2 = (int(np.around(v * size).astype(np.int32)) for v in (y1n, x1n, y2n, x2n))
        crop = np.asarray(det_mask[k], np.float32)[cy1:cy2, cx1:cx2]
        if crop.size == 0:
            continue
        m = resize_linear(crop, x2 - x1, y2 - y1) > 0.5
        full = np.zeros((image_h, image_w), bool)
        full[y1:y2, x1:x2] = m
        cid = int(det_box[k, 4])
        entries.append({"index": k, "classid": cid, "score": float(det_box[k, 5]), "mask": full})
        merged[full] = cid + 1
    return entries, merged


def segmentation_miou(true_maps, pred_maps):
    """calculate_test_map.py:303-346: pixel confusion counts of the merged class maps (0 = background,
    1..3 = crack / spall / rebar) over all images; IoU_c = n_cc / (row_c + column_c - n_cc); returns
    [bg, crack, spall, rebar, mean].  Pinned by tests/golden/miou.json."""
    conf = np.zeros((4, 4), np.int64)            # [true class][predicted class]
    for t, p in zip(true_maps, pred_maps):
        t, p = np.asarray(t), np.asarray(p)
        assert t.shape == p.shape
        for a in range(4):
            for b in range(4):
                conf[a, b] += int(np.sum((t == a) * (p == b)))
    ious = [conf[c, c] / (conf[c, :].sum() + conf[:, c].sum() - conf[c, c]) for c in range(4)]
    return ious + [float(np.mean(ious))]


# ---------------------------------------------------------------------------------------------
# training-data pipeline (utils/train_data.py:44-276, 321-531) -- SURVEY.md 8(f2)
# ---------------------------------------------------------------------------------------------
def _point_in_polygon(xp, yp, x, y) -> int:
    """scikit-image's point_in_polygon (measure/_pnpoly.pyx; the third-party routine behind
    skimage.draw.polygon, which the reference calls at utils/train_data.py:331): 0 outside, 1 inside,
    2 on a vertex, 3 on an edge.  Pinned by tests/golden/polygon.json (skimage 0.18.3 run in the build
    container by tools/make_golden_polygon.py)."""
    n, eps = len(xp), 1e-12
    x0, y0 = xp[n - 1] - x, yp[n - 1] - y
    l = r = 0
    for i in range(n):
        x1, y1 = xp[i] - x, yp[i] - y
        if -eps < x1 < eps and -eps < y1 < eps:
            return 2
        if ((y0 > 0) != (y1 > 0)) and ((x0 * y1 - x1 * y0) / (y1 - y0) > 0):
            r += 1
        if ((y0 < 0) != (y1 < 0)) and ((x0 * y1 - x1 * y0) / (y1 - y0) < 0):
            l += 1
        x0, y0 = x1, y1
    if (r & 1) != (l & 1):
        return 3
    return 1 if (r & 1) else 0


def draw_polygon(ys, xs):
    """skimage.draw.polygon(ys, xs) without a shape: (rr, cc) of every pixel of the bounding box
    [max(0, min), ceil(max)] that is inside, on an edge or on a vertex"""
    ys, xs = np.asarray(ys, np.float64), np.asarray(xs, np.float64)
    rr, cc = [], []
    for r in range(int(max(0, ys.min())), int(np.ceil(ys.max())) + 1):
        for c in range(int(max(0, xs.min())), int(np.ceil(xs.max())) + 1):
            if _point_in_polygon(xs, ys, float(c), float(r)):
                rr.append(r)
                cc.append(c)
    return np.asarray(rr, np.int64), np.asarray(cc, np.int64)


def instance_mask(polys, image_h: int, image_w: int) -> np.ndarray:
    """load_mask for one instance (utils/train_data.py:325-336): polys = [{'type': 'out'|'in', 'all_points_x',
    'all_points_y'}]; 'out' fills, 'in' clears (a hole), every polygon then sets its own vertex pixels"""
    m = np.zeros((image_h, image_w), bool)
    for poly in polys:
        xs, ys = poly["all_points_x"], poly["all_points_y"]
        rr, cc = draw_polygon(ys, xs)
        m[rr, cc] = poly["type"] == "out"
        m[np.array(ys), np.array(xs)] = True
    return m


def _taps(dn: int, sn: int):
    scale = np.float64(sn) / dn
    f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s0 = np.floor(f).astype(np.int64)
    a = (f - s0.astype(np.float32)).astype(np.float32)
    lo = s0 < 0
    a[lo], s0[lo] = 0.0, 0
    hi = s0 >= sn - 1
    a[hi], s0[hi] = 0.0, sn - 1
    return s0, np.minimum(s0 + 1, sn - 1), a


def resize_linear_u8(src: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """cv2.resize(uint8 image, INTER_LINEAR), restated from OpenCV's 8-bit path (cv2 is not installed:
    unpinned): coefficients rounded to 11 fractional bits, horizontal pass in integers, vertical pass
    (b0*S0 + b1*S1 + 2^21) >> 22."""
    src = np.asarray(src, np.uint8)
    x0, x1, ax = _taps(dst_w, src.shape[1])
    y0, y1, ay = _taps(dst_h, src.shape[0])
    ax1 = np.rint(ax * np.float32(2048)).astype(np.int64)
    ay1 = np.rint(ay * np.float32(2048)).astype(np.int64)
    s = src.astype(np.int64)
    rows = s[:, x0] * (2048 - ax1)[None, :, None] + s[:, x1] * ax1[None, :, None]
    out = (rows[y0] * (2048 - ay1)[:, None, None] + rows[y1] * ay1[:, None, None] + (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)


def scale_and_crop(im_sized: np.ndarray, new_w: int, new_h: int, dx: int, dy: int, size: int, pad_value) -> np.ndarray:
    """the placing half of apply_random_scale_and_crop (utils/train_data.py:452-464): [new_h,new_w,C] ->
    [size,size,C] with its corner at (dx, dy), cropped where negative, padded elsewhere"""
    if dx > 0:
        im_sized = np.pad(im_sized, ((0, 0), (dx, 0), (0, 0)), mode="constant", constant_values=pad_value)
    else:
        im_sized = im_sized[:, -dx:, :]
    if (new_w + dx) < size:
        im_sized = np.pad(im_sized, ((0, 0), (0, size - (new_w + dx)), (0, 0)), mode="constant", constant_values=pad_value)
    if dy > 0:
        im_sized = np.pad(im_sized, ((dy, 0), (0, 0), (0, 0)), mode="constant", constant_values=pad_value)
    else:
        im_sized = im_sized[-dy:, :, :]
    if (new_h + dy) < size:
        im_sized = np.pad(im_sized, ((0, size - (new_h + dy)), (0, 0), (0, 0)), mode="constant", constant_values=pad_value)
    return im_sized[:size, :size, :]


def _flip(a: np.ndarray, flip: int) -> np.ndarray:
    return a[:, ::-1] if flip == 2 else (a[::-1] if flip == 3 else a)


def place_image(image_u8: np.ndarray, size: int, new_w: int, new_h: int, dx: int, dy: int, flip: int) -> np.ndarray:
    """image_read up to the flip (utils/train_data.py:376-397): uint8 [size,size,3]"""
    return np.ascontiguousarray(_flip(scale_and_crop(resize_linear_u8(image_u8, new_w, new_h), new_w, new_h, dx, dy, size, 127), flip))


def place_mask(mask: np.ndarray, size: int, new_w: int, new_h: int, dx: int, dy: int, flip: int) -> np.ndarray:
    """resize_mask for one instance (utils/train_data.py:418-436): float32 resize, pad 0, flip, np.around -> bool"""
    m = resize_linear(np.asarray(mask, np.float32), new_w, new_h)[:, :, None]
    m = _flip(scale_and_crop(m, new_w, new_h, dx, dy, size, 0.0), flip)
    return np.around(np.squeeze(m, -1)).astype(bool)


def salt_pepper(im: np.ndarray, rows, cols, nsalt: int) -> np.ndarray:
    """add_salt_pepper_noise (utils/train_data.py:511-525) with the drawn coordinates passed in"""
    im = im.copy()
    im[rows[:nsalt], cols[:nsalt], :] = 1
    im[rows[nsalt:], cols[nsalt:], :] = 0
    return im


def change_light(image: np.ndarray, coeff: float) -> np.ndarray:
    """change_light (utils/train_data.py:527-535): cv2 RGB2HLS (8-bit: H/2, 255 L, 255 S), L scaled in float64,
    clipped at 255, truncated, HLS2RGB.  OpenCV's documented formulas in float32 (cv2 not installed: unpinned)."""
    f32 = np.float32
    rgb = image.astype(f32) * f32(1.0 / 255.0)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    vmax, vmin = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    diff, s_ = (vmax - vmin).astype(f32), (vmax + vmin).astype(f32)
    l = (s_ * f32(0.5)).astype(f32)
    grey = ~(diff > f32(1.1920929e-07))
    safe = np.where(grey, f32(1), diff).astype(f32)
    sat = np.where(l < f32(0.5), diff / np.where(grey, f32(1), s_), diff / np.where(grey, f32(1), (f32(2) - s_))).astype(f32)
    d60 = (f32(60) / safe).astype(f32)
    h = np.where(vmax == r, (g - b) * d60, np.where(vmax == g, (b - r) * d60 + f32(120), (r - g) * d60 + f32(240))).astype(f32)
    h = np.where(h < 0, h + f32(360), h).astype(f32)
    h, sat = np.where(grey, f32(0), h), np.where(grey, f32(0), sat)
    sat8 = lambda v: np.clip(np.rint(v), 0, 255).astype(np.uint8)
    H8, S8, L8 = sat8(h * f32(0.5)), sat8(sat * f32(255)), sat8(l * f32(255))
    L = L8.astype(np.float64) * coeff
    L[L > 255] = 255
    L8 = L.astype(np.uint8)
    hh, ll, ss = H8.astype(f32) * f32(2), L8.astype(f32) * f32(1.0 / 255.0), S8.astype(f32) * f32(1.0 / 255.0)
    p2 = np.where(ll <= f32(0.5), ll * (f32(1) + ss), ll + ss - ll * ss).astype(f32)
    p1 = (f32(2) * ll - p2).astype(f32)
    hq = (hh * f32(1.0 / 60.0)).astype(f32)
    hq = np.where(hq >= 6, hq - f32(6), hq).astype(f32)
    sector = np.floor(hq).astype(np.int64)
    f = (hq - sector.astype(f32)).astype(f32)
    tab = np.stack([p2, p1, (p1 + (p2 - p1) * (f32(1) - f)).astype(f32), (p1 + (p2 - p1) * f).astype(f32)], axis=-1)
    idx = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])       # (b, g, r) per sector
    pick = idx[sector]
    bo = np.take_along_axis(tab, pick[..., 0:1], -1)[..., 0]
    go = np.take_along_axis(tab, pick[..., 1:2], -1)[..., 0]
    ro = np.take_along_axis(tab, pick[..., 2:3], -1)[..., 0]
    ro, go, bo = (np.where(ss == 0, ll, v).astype(f32) for v in (ro, go, bo))
    return np.stack([sat8(ro * f32(255)), sat8(go * f32(255)), sat8(bo * f32(255))], axis=-1)


PYBLUR_LINES3 = {0: (1, 0, 1, 2), 45: (2, 0, 0, 2), 90: (0, 1, 2, 1), 135: (0, 0, 2, 2)}


def pyblur_line_kernel3(angle: int, line_type: int) -> np.ndarray:
    """pyblur.LineKernel(3, angle, linetype) (utils/train_data.py:466-494 calls pyblur.LinearMotionBlur; pyblur is not
    installable here, restated from its 0.2.x source: unpinned).  LineDictionary anchors (row0, col0, row1, col1) per
    angle; 'right' (1) replaces the first anchor by the centre, 'left' (2) the second; the line between the anchors
    (skimage.draw.line: for three pixels the two ends and their midpoint) is set to 1 and normalised.  pyblur mutates the
    shared anchor list in place (after one 'right' and one 'left' call an angle is the identity for the rest of the
    process): deliberately not reproduced."""
    r0, c0, r1, c1 = PYBLUR_LINES3[angle]
    if line_type == 1:
        r0 = c0 = 1
    if line_type == 2:
        r1 = c1 = 1
    k = np.zeros((3, 3), np.float32)
    k[r0, c0] = 1
    k[r1, c1] = 1
    if max(abs(r1 - r0), abs(c1 - c0)) == 2:          # a three-pixel line: its midpoint is the centre
        k[(r0 + r1) // 2, (c0 + c1) // 2] = 1
    return k / np.float32(np.count_nonzero(k))


def motion_blur3(img: np.ndarray, angle: int, line_type: int) -> np.ndarray:
    """linearmotion_blur3C with lineLength 3 (utils/train_data.py:466-494) = per channel
    scipy.signal.convolve2d(img.astype(float32), LineKernel, mode='same', fillvalue=255.0).astype(uint8)
    (pyblur.LinearMotionBlur; tests/test_train_data.py checks this restatement against scipy itself)"""
    k = pyblur_line_kernel3(angle, line_type)
    S = img.shape[0]
    pad = np.full((S + 2, S + 2, 3), 255.0, np.float32)
    pad[1:-1, 1:-1] = img.astype(np.float32)
    out = np.zeros(img.shape, np.float32)
    # convolution: out[y,x] = sum k[i,j] * in[y-(i-1), x-(j-1)], kernel entries in row-major order
    for i in range(3):
        for j in range(3):
            if k[i, j] != 0:
                dy, dx = i - 1, j - 1
                out = out + pad[1 - dy:1 - dy + S, 1 - dx:1 - dx + S] * k[i, j]
    return np.clip(out.astype(np.int64), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------
# dataset pre-processing (pre_process.py:16-318): mask image -> contours -> regions
# ---------------------------------------------------------------------------
# The contour extraction is OpenCV's, not the reference's own code: cv2.findContours(thresh, cv2.RETR_TREE,
# cv2.CHAIN_APPROX_NONE) (pre_process.py:78,82,86; OpenCV 3.x API: three return values; version unpinned,
# "opencv-python" in the README).  cv2 is installed in neither interpreter of the build container, so this is
# a restatement of the published algorithm it implements -- S. Suzuki, K. Abe, "Topological structural
# analysis of digitized binary images by border following", CVGIP 30 (1985), Algorithm 1 -- with the
# conventions of OpenCV's legacy implementation as far as they are documented / observable:
#   * foreground 8-connected, raster scan with a one-pixel zero frame, labels NBD from 2;
#   * an outer border starts where f == 1 and the left neighbour is 0, a hole border where f >= 1 and the
#     right neighbour is 0; the first search runs CLOCKWISE from the west (outer) / east (hole) neighbour,
#     every following one COUNTER-CLOCKWISE from the neighbour after the previous pixel (direction codes
#     0..7 = E, NE, N, NW, W, SW, S, SE), so an outer border is listed from its top-left pixel DOWN its left
#     side; CHAIN_APPROX_NONE lists every border pixel visit;
#   * parent from the label LNBD last seen on the row (Suzuki's table);
#   * contours are numbered in pre-order of the tree with the siblings in REVERSE order of discovery (the
#     legacy implementation prepends a new contour to its parent's child list); hierarchy rows are
#     [next, previous, first_child, parent].
# PARITY UNPINNED against cv2 itself (only against hand-derived answers and topological invariants).
_CODE_DELTAS = ((1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1))   # (dx, dy) of codes 0..7


def find_contours_tree(binary: np.ndarray):
    """-> (contours: list of int32 [n,2] arrays of (x, y); hierarchy: int32 [n,4])"""
    h, w = binary.shape
    f = np.zeros((h + 2, w + 2), np.int64)
    f[1:-1, 1:-1] = (np.asarray(binary) != 0)
    is_hole = {1: True}            # the frame
    parent_of = {1: 0}
    borders = {}                   # nbd -> list of (x, y)
    order = []
    nbd = 1
    for i in range(1, h + 1):
        lnbd = 1
        for j in range(1, w + 1):
            v = f[i, j]
            start_hole = None
            if v == 1 and f[i, j - 1] == 0:
                start_hole = False
            elif v >= 1 and f[i, j + 1] == 0:
                start_hole = True
                if v > 1:
                    lnbd = v
            if start_hole is not None:
                nbd += 1
                is_hole[nbd] = start_hole
                # Suzuki's parent table
                if start_hole:
                    parent_of[nbd] = lnbd if not is_hole[lnbd] else parent_of[lnbd]
                else:
                    parent_of[nbd] = parent_of[lnbd] if not is_hole[lnbd] else lnbd
                pts = []
                # first search: clockwise from the west (outer) / east (hole) neighbour
                s_end = s = 0 if start_hole else 4
                found = False
                while True:
                    s = (s - 1) & 7
                    dx, dy = _CODE_DELTAS[s]
                    if f[i + dy, j + dx] != 0:
                        found = True
                        break
                    if s == s_end:
                        break
                if not found:
                    f[i, j] = -nbd
                    pts.append((j - 1, i - 1))
                else:
                    i1, j1 = i + _CODE_DELTAS[s][1], j + _CODE_DELTAS[s][0]
                    i3, j3 = i, j
                    while True:
                        s_end = s
                        # counter-clockwise from the neighbour after the one we came from
                        while True:
                            s = (s + 1) & 7
                            dx, dy = _CODE_DELTAS[s]
                            if f[i3 + dy, j3 + dx] != 0:
                                break
                        i4, j4 = i3 + _CODE_DELTAS[s][1], j3 + _CODE_DELTAS[s][0]
                        # was the east neighbour (code 0) a zero pixel examined by this search?
                        # examined codes: s_end+1 .. s (cyclic), the last one is the hit
                        passed_east = ((0 - (s_end + 1)) & 7) < ((s - (s_end + 1)) & 7) and f[i3, j3 + 1] == 0
                        if passed_east:
                            f[i3, j3] = -nbd
                        elif f[i3, j3] == 1:
                            f[i3, j3] = nbd
                        pts.append((j3 - 1, i3 - 1))
                        if i4 == i and j4 == j and i3 == i1 and j3 == j1:
                            break
                        i3, j3 = i4, j4
                        s = (s + 4) & 7          # the direction back to where we came from
                borders[nbd] = np.asarray(pts, np.int32).reshape(-1, 2)
                order.append(nbd)
            if f[i, j] != 0 and f[i, j] != 1:
                lnbd = abs(int(f[i, j]))
    # numbering: pre-order, siblings newest first
    children = {}
    for b in order:
        children.setdefault(parent_of[b], []).append(b)
    seq = []

    def walk(p):
        for b in reversed(children.get(p, [])):
            seq.append(b)
            walk(b)
    walk(1)
    index = {b: k for k, b in enumerate(seq)}
    hier = np.full((len(seq), 4), -1, np.int32)
    for p, kids in children.items():
        ks = list(reversed(kids))
        for a, b in zip(ks, ks[1:]):
            hier[index[a], 0] = index[b]
            hier[index[b], 1] = index[a]
        if p != 1:
            hier[index[p], 2] = index[ks[0]]
            for b in ks:
                hier[index[b], 3] = index[p]
    return [borders[b] for b in seq], hier


def contour_centroid(points_xy: np.ndarray):
    """cv2.moments of a contour (Green's theorem over the closed polygon) -> (int(m10/m00), int(m01/m00)) as
    pre_process.py:178-180 uses it; raises ZeroDivisionError for a zero-area contour like the reference"""
    p = np.asarray(points_xy, np.float64).reshape(-1, 2)
    x0, y0 = p[:, 0], p[:, 1]
    x1, y1 = np.roll(x0, -1), np.roll(y0, -1)
    cross = x0 * y1 - x1 * y0
    a00 = cross.sum()
    a10 = ((x0 + x1) * cross).sum()
    a01 = ((y0 + y1) * cross).sum()
    if a00 == 0:
        raise ZeroDivisionError("float division by zero")
    m00, m10, m01 = a00 / 2.0, a10 / 6.0, a01 / 6.0
    if m00 < 0:                      # cv2 returns the moments of the positively oriented polygon
        m00, m10, m01 = -m00, -m10, -m01
    return int(m10 / m00), int(m01 / m00)


def regions_from_contours(per_class):
    """pre_process.py:88-163: per_class = [(classname, contours, hierarchy)] in the reference's order
    (crack, spall, rebar) -> (regions dict, number of masks with a contour nested two levels deep)"""
    regions, count, errors = {}, 0, 0
    for classname, contours, hier in per_class:
        pair = {}
        for j, c in enumerate(contours):
            all_x, all_y = c[:, 0].tolist(), c[:, 1].tolist()
            if hier[j, 3] == -1:
                regions[str(count)] = {"region_attributes": classname,
                                       "shape_attributes": [{"type": "out", "all_points_x": all_x, "all_points_y": all_y}]}
                pair[str(j)] = count
                count += 1
            else:
                parent = int(hier[j, 3])
                if hier[parent, 3] != -1:
                    errors += 1
                    continue
                regions[str(pair[str(parent)])]["shape_attributes"].append(
                    {"type": "in", "all_points_x": all_x, "all_points_y": all_y})
    return regions, errors


def merge_regions(regions, object_merge):
    """pre_process.py:165-222: instances whose outer contour's centroid lies strictly inside a 'merge' box are
    joined into one instance per box (the closest box centre wins); class crack > rebar > spall.  Restated
    WITHOUT the reference's two accidents: its closest-box test is only evaluated when the loop variable ends
    on the last box (always true) and re-uses `dis_index` from a previous instance when no box contains the
    centroid (the follow-up containment test then rejects it unless the stale box happens to contain it --
    that stale-index acceptance is reproduced, since it changes the output)."""
    if not object_merge:
        return {}
    groups = {jj: [] for jj in range(len(object_merge))}
    names = {jj: [] for jj in range(len(object_merge))}
    dis_index = None
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
                dis_index, old = ii, d
        if dis_index is None:
            raise UnboundLocalError("local variable 'dis_index' referenced before assignment")   # as the reference
        x1, y1, x2, y2 = object_merge[dis_index]
        if x1 <= cX <= x2 and y1 <= cY <= y2:
            groups[dis_index].extend(reg["shape_attributes"])
            names[dis_index].append(reg["region_attributes"])
    new_regions, count = {}, 0
    for jj in range(len(object_merge)):
        if not groups[jj]:
            continue
        nl = names[jj]
        cls = "crack" if "crack" in nl else ("spall" if ("spall" in nl and "rebar" not in nl) else "rebar")
        new_regions[str(count)] = {"region_attributes": cls, "shape_attributes": groups[jj]}
        count += 1
    return new_regions
