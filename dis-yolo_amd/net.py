"""Host-side mirror of the reference's ``YOLONet`` (yolo/yolo3_net_pos.py:12-65).

The TF-1.x graph object becomes a plan over the gfx950 kernel library: the 82-layer
topology (yolo/yolo3_net_pos.py:159-412) is a data table, every layer is one or two
kernel launches through the C ABI (``lib.py``), and PyTorch only owns device memory,
streams and (for data parallel training) the RCCL process group.

Same constructor / attribute names as the reference where they exist:
``YOLONet(training)``, ``batchsize``, ``classes``, ``num_class``, ``anchors``,
``num_anchor``, ``output_depth``, ``k``, ``k_mapout``, ``*_scale``.  ``sess.run``
fetches become methods: ``forward`` (= ``logits``), ``evaluation``, ``train_step``
(= ``[total_loss, optimizer]``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import math
import json
import contextlib
import os
import sys
import numpy as np
import torch

from . import config as cfg
from . import lib as L

BF16 = torch.bfloat16
F32 = torch.float32



class Layer:
    __slots__ = ("idx", "cin", "cout", "k", "stride", "kind", "src", "src_up", "shortcut", "lock", "H", "W", "Ho",
                 "Wo", "w", "bias", "gamma", "beta", "mm", "mv", "scale", "shift", "mean", "rstd", "wp", "wdg", "raw",
                 "act", "stats", "stats_rows", "desc", "grad", "grad_set", "need_grad", "dw", "dbias", "dgamma",
                 "dbeta", "dx", "pad_t", "pad_l", "dgrad_descs", "wgrad_desc", "cout_pad",
                 "act8", "w8", "s_w", "s_out", "escale", "desc8", "dual16", "bn_sums", "bwd_local", "bwd_global", "bwd_part", "bwd_part_rows", "wq",
                 "csync", "fused_fwd", "csync_bwd", "fused_bwd")

    def __init__(self, idx, cin, cout, k, stride, kind, src, src_up=None, shortcut=None):
        self.idx, self.cin, self.cout, self.k, self.stride, self.kind = idx, cin, cout, k, stride, kind
        self.src, self.src_up, self.shortcut = src, src_up, shortcut
        for s in self.__slots__[9:]:
            setattr(self, s, None)


def build_topology(num_class: int, k_map: int) -> List[Layer]:
    """The layer table of build_network (yolo/yolo3_net_pos.py:159-412, active m=1/2 mask
    subnet).  ``src`` = producing layer (0 = the image); ``src_up`` = layer whose output is
    nearest-upsampled x2 and concatenated AFTER ``src`` (:290-291); ``shortcut`` = layer
    added after the activation (:148-151)."""
    out_depth = (num_class + 5) * 3
    Ls: List[Layer] = []

    def add(idx, cin, cout, k, s, kind, src, src_up=None, shortcut=None):
        Ls.append(Layer(idx, cin, cout, k, s, kind, src, src_up, shortcut))

    add(1, 3, 32, 3, 1, "bn", 0)
    add(2, 32, 64, 3, 2, "bn", 1)
    i = 3
    ch = 64
    for nblocks, down in ((1, 5), (2, 10), (8, 27), (8, 44), (4, None)):
        for _ in range(nblocks):
            add(i, ch, ch // 2, 1, 1, "bn", i - 1)
            add(i + 1, ch // 2, ch, 3, 1, "res", i, shortcut=i - 1)
            i += 2
        if down:
            add(down, ch, ch * 2, 3, 2, "bn", down - 1)
            ch *= 2
            i = down + 1
    # head 1
    add(53, 1024, 512, 1, 1, "bn", 52)
    add(54, 512, 1024, 3, 1, "bn", 53)
    add(55, 1024, 512, 1, 1, "bn", 54)
    add(56, 512, 1024, 3, 1, "bn", 55)
    add(57, 1024, 512, 1, 1, "bn", 56)
    add(58, 512, 1024, 3, 1, "bn", 57)
    add(59, 1024, out_depth, 1, 1, "lin", 58)
    # head 2
    add(60, 512, 256, 1, 1, "bn", 57)
    add(61, 768, 256, 1, 1, "bn", 43, src_up=60)
    add(62, 256, 512, 3, 1, "bn", 61)
    add(63, 512, 256, 1, 1, "bn", 62)
    add(64, 256, 512, 3, 1, "bn", 63)
    add(65, 512, 256, 1, 1, "bn", 64)
    add(66, 256, 512, 3, 1, "bn", 65)
    add(67, 512, out_depth, 1, 1, "lin", 66)
    # head 3
    add(68, 256, 128, 1, 1, "bn", 65)
    add(69, 384, 128, 1, 1, "bn", 26, src_up=68)
    add(70, 128, 256, 3, 1, "bn", 69)
    add(71, 256, 128, 1, 1, "bn", 70)
    add(72, 128, 256, 3, 1, "bn", 71)
    add(73, 256, 128, 1, 1, "bn", 72)
    add(74, 128, 256, 3, 1, "bn", 73)
    add(75, 256, out_depth, 1, 1, "lin", 74)
    # mask subnet m = 1/2
    add(76, 128, 64, 1, 1, "bn", 73)
    add(77, 192, 64, 1, 1, "bn", 9, src_up=76)
    add(78, 64, 128, 3, 1, "bn", 77)
    add(79, 128, 32, 1, 1, "bn", 78)
    add(80, 96, 32, 1, 1, "bn", 4, src_up=79)
    add(81, 32, 64, 3, 1, "bn", 80)
    add(82, 64, k_map * k_map, 1, 1, "lin", 81)
    Ls.sort(key=lambda l: l.idx)
    # the residual shortcut of block [1x1 (i-1), 3x3 res (i)] is the block input = out(i-2)
    for l in Ls:
        if l.kind == "res":
            l.shortcut = l.idx - 2
    assert [l.idx for l in Ls] == list(range(1, 83))
    return Ls


def var_name(i: int, leaf: str) -> str:
    """checkpoint variable names (train_yolo3_mask.py:86-103)"""
    return "yolo/convolutional%d/%s" % (i, leaf)


class YOLONet(object):
    def __init__(self, training: bool = False, device=None, image_size: Optional[int] = None,
                 batch_size: Optional[int] = None, stage: int = 1, lock: Optional[Dict[int, bool]] = None,
                 seed: int = 0, xavier_locked: bool = True, plan_only: bool = False, dtype: str = "bf16",
                 backbone_pair: bool = False):
        # 1. parameters (yolo/yolo3_net_pos.py:15-38)
        self.batchsize = int(batch_size if batch_size is not None else cfg.BATCH_SIZE)
        self.classes = cfg.CLASSES
        self.num_class = len(self.classes)
        self.anchors = np.asarray(cfg.ANCHORS, dtype=np.float32)
        self.num_anchor = 3
        self.output_depth = (self.num_class + 5) * self.num_anchor
        self.k = cfg.K_MAP
        self.k_mapout = self.k * self.k
        self.object_scale = cfg.OBJECT_SCALE
        self.noobject_scale = cfg.NOOBJECT_SCALE
        self.class_scale = cfg.CLASS_SCALE
        self.coord_scale = cfg.COORD_SCALE
        self.mask_scale = cfg.MASK_SCALE
        self.l2 = cfg.L2_WEIGHT
        self.training = bool(training)
        # "fp8": the inference-mode layers conv1-52 (the locked backbone of stage 1; every call of an
        # inference net) keep their activations and MFMA operands in OCP e4m3 with per-tensor scales
        # (calibrate_fp8()); everything else stays bf16.  BASELINE.json configs[4].
        if dtype not in ("bf16", "fp8"):
            raise ValueError("dtype must be 'bf16' or 'fp8'")
        self.dtype = dtype
        self.fp8_ready = False
        self.image_size = int(image_size if image_size is not None else cfg.IMAGE_SIZE)
        if self.image_size % 32:
            raise ValueError("image size must be a multiple of 32")
        # plan_only: variables, parameter arena and layer table only (any device, no kernel library) --
        # what checkpoint tools and the CPU tests of the data-parallel protocol need; it cannot compute
        self.plan_only = bool(plan_only)
        if device is None:
            if self.plan_only:
                device = torch.device("cpu")
            elif not torch.cuda.is_available():
                raise L.DisyoloError("YOLONet needs a GPU: the HIP kernels have no CPU fallback")
            else:
                device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if not self.plan_only:
            L.load()
            self.ws = L.Workspace(self.device)
            self.ws_aux = L.Workspace(self.device)      # scratch of the side lane (weight gradients)
            self.ws_det = L.Workspace(self.device)      # scratch of the detection filter (either lane)
        self._reg_fresh = False
        # lock map: stage 1 = conv1-52 locked (shipped source), stage 2 = all trainable
        self.lock = dict(lock) if lock is not None else {i: (stage == 1 and i <= 52) for i in range(1, 83)}
        self.layers = build_topology(self.num_class, self.k)
        self.by_idx = {l.idx: l for l in self.layers}
        self._lr = float(cfg.LEARNING_RATE)
        self.lr_dev = None       # device copy read by the optimizer kernel (set in _init_params)
        self.dp = None  # set by enable_data_parallel
        self.sync_bn = False    # enable_data_parallel(sync_bn=True): batch statistics over all ranks
        self._rec = None        # (list, cut points) while a step is being recorded
        self._prog = None       # recorded command list of one training step
        self._prog_marks = []   # [(command index, layer)] all-reduce trigger points
        self._graph = None      # hipGraph of the recorded step (single GPU)
        self.opt_chunks = None      # slices of the arena the optimizer sweeps one by one (_plan_opt_chunks)
        # build_program(overlap_tail=True): the recorded step does not join its side lane at the end -- the optimizer's last
        # sweeps and the last weight gradients keep running while the NEXT replay's locked-backbone forward starts; the
        # next replay waits, tensor by tensor, for the side-lane work that still reads it (cmdlist slots)
        self._overlap = False
        self._overlap_rec = False      # while such a step is being recorded
        self._dp_pending = []          # data parallel, exchange in the list: slices exchanged and not yet swept
        self._tail_open = False        # a replay's side lane may still be running: join before anything but the next replay
        self._progs = None      # [parity] -> (list, marks, bwd_end) of the pipelined step
        # conv1 + conv2 as ONE launch wherever both run in inference mode (the locked backbone of stage 1, every inference
        # net): conv1's output -- 170 MB at B = 8, one consumer, unused by the active mask subnet (yolo/yolo3_net_pos.py:163)
        # -- is then never written, and ``by_idx[1].act`` is NOT filled.  Set to False to get every layer's output.
        self.fuse_first_two = os.environ.get("DISYOLO_FUSE12", "1") != "0"
        self.fuse_blocks = os.environ.get("DISYOLO_FUSE_BLOCKS", "1") != "0"
        self.use_side_lane = os.environ.get("DISYOLO_SIDE_LANE", "1") != "0"
        # training-mode batch norm INSIDE the conv launches of the main lane (DISYOLO_CONV_BN_FUSED, csrc/conv_common.h
        # "cluster exchange"): wherever a trainable layer's whole grid is resident at once, its bn_finalize + bn_act_fwd
        # launches (forward) disappear into the conv's epilogue.  Off: DISYOLO_BN_INKERNEL=0, or bn_inkernel = False before
        # the descriptors are (re)built.  Never on the side lane's layers: two such launches must not run concurrently.
        self.bn_inkernel = os.environ.get("DISYOLO_BN_INKERNEL", "1") != "0"
        # ... and the backward form (DISYOLO_CONV_BN_BWD_FUSED): the data-gradient conv that makes a layer's output gradient
        # final runs that layer's whole batch-norm backward in its epilogue (colreduce + bn_bwd_finalize + bn_bwd_apply gone)
        # Built, parity-green, measured and left OFF (DISYOLO_BN_INKERNEL_BWD=1 turns it on): with the side lane's weight
        # gradients beside it the backward pass is bound by what the two lanes' kernels take from the CUs, not by the main
        # lane's launch count -- 14 layers x 3 launches fewer changed the step by -1.2 ... +0.2 % (profiles/r06_bn_inkernel.txt)
        self.bn_inkernel_bwd = os.environ.get("DISYOLO_BN_INKERNEL_BWD", "0") == "1"
        # debug mode of the fused batch-norm backward (set to a list): every layer whose backward sums come from the
        # data-gradient conv's epilogue ALSO runs the plain column reduction on the same gradient (eager steps only) and
        # appends the relative differences of dx / dgamma / dbeta
        self.bn_fuse_check = None
        if os.environ.get("DISYOLO_EXP_SKIP_WGRAD") in ("1", "2"):
            print("disyolo: DISYOLO_EXP_SKIP_WGRAD is set -- weight gradients are NOT computed (timing experiment)", file=sys.stderr)
        # weight gradients of the last layers of the backward pass stay on the main lane (tuned below)
        self.tail_on_main = int(os.environ.get("DISYOLO_TAIL_MAIN", "0"))
        # backbone_pair (stage 1): the locked backbone -- whose output depends on nothing a step updates -- runs ONCE
        # per TWO batches at batch size 2B (a 576^2 B = 8 layer is a one-round grid: 169 us per image at B = 8, 137 at
        # B = 16), the trainable part steps through the two halves one after the other.  Every image still passes
        # every layer exactly once; the weights after N steps are those of N plain steps up to the f32 summation order
        # of the differently tiled backbone kernels.  set_batch(batch, half), build_program(pair=True).
        self.pair = bool(backbone_pair)
        self._half = 0
        self._pair_P = 0
        self._init_params(seed, xavier_locked)
        if self.pair:
            if not self.training or self.dtype != "bf16" or self.plan_only:
                raise L.DisyoloError("backbone_pair needs a bf16 training net")
            self._pair_P = self._backbone_prefix()
            if self._pair_P < 2:
                raise L.DisyoloError("backbone_pair needs a locked layer prefix (stage 1)")
        if not self.plan_only:
            self._plan(self.batchsize, self.image_size)

    # ------------------------------------------------------------------ parameters
    def _init_params(self, seed: int, xavier_locked: bool) -> None:
        """Variables with the reference initialisers (yolo/yolo3_net_pos.py:77-86,111-123,
        134-140).  Trainable ones live in one flat f32 arena [weights+biases | gamma+beta] so
        Adam and the gradient all-reduce are single contiguous sweeps."""
        dev = self.device
        g = torch.Generator().manual_seed(seed)
        n_decay = 0
        n_nodecay = 0
        for l in self.layers:
            l.lock = self.lock[l.idx]
            if not l.lock:
                n_decay += l.k * l.k * l.cin * l.cout + (l.cout if l.kind == "lin" else 0)
                if l.kind != "lin":
                    n_nodecay += 2 * l.cout
        self.n_decay, self.n_params = n_decay, n_decay + n_nodecay
        n = max(self.n_params, 1)
        self.arena = torch.zeros(n, dtype=F32, device=dev)
        self.grad_arena = torch.zeros(n, dtype=F32, device=dev)
        self.adam_m = torch.zeros(n, dtype=F32, device=dev)
        self.adam_v = torch.zeros(n, dtype=F32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)   # Adam's t, device resident
        self.lr_dev = torch.full((1,), self._lr, dtype=F32, device=dev)
        self.arena_slices: Dict[str, Tuple[int, int]] = {}
        off_d, off_n = 0, n_decay
        self.params: Dict[str, torch.Tensor] = {}

        def take(name, shape, region):
            nonlocal off_d, off_n
            cnt = int(np.prod(shape))
            if region == "decay":
                o = off_d
                off_d += cnt
            else:
                o = off_n
                off_n += cnt
            self.arena_slices[name] = (o, cnt)
            return self.arena[o:o + cnt].view(shape), self.grad_arena[o:o + cnt].view(shape)

        for l in self.layers:
            shape = (l.k, l.k, l.cin, l.cout)
            if l.lock and not xavier_locked:
                w = torch.empty(shape, dtype=torch.float64)
                torch.nn.init.trunc_normal_(w, 0.0, 0.001, -0.002, 0.002, generator=g)
            else:
                lim = math.sqrt(6.0 / (l.k * l.k * l.cin + l.k * l.k * l.cout))
                w = (torch.rand(shape, dtype=torch.float64, generator=g) * 2 - 1) * lim
            w = w.to(F32)
            if l.lock:
                l.w = w.to(dev)
            else:
                l.w, l.dw = take(var_name(l.idx, "weights"), shape, "decay")
                l.w.copy_(w)
            self.params[var_name(l.idx, "weights")] = l.w
            if l.kind == "lin":
                if l.lock:
                    l.bias = torch.zeros(l.cout, dtype=F32, device=dev)
                else:
                    l.bias, l.dbias = take(var_name(l.idx, "biases"), (l.cout,), "decay")
                self.params[var_name(l.idx, "biases")] = l.bias
            else:
                if l.lock:
                    l.gamma = torch.ones(l.cout, dtype=F32, device=dev)
                    l.beta = torch.zeros(l.cout, dtype=F32, device=dev)
                else:
                    l.gamma, l.dgamma = take(var_name(l.idx, "BatchNorm/gamma"), (l.cout,), "nodecay")
                    l.beta, l.dbeta = take(var_name(l.idx, "BatchNorm/beta"), (l.cout,), "nodecay")
                    l.gamma.fill_(1.0)
                l.mm = torch.zeros(l.cout, dtype=F32, device=dev)
                l.mv = torch.ones(l.cout, dtype=F32, device=dev)
                for leaf, t in (("gamma", l.gamma), ("beta", l.beta), ("moving_mean", l.mm),
                                ("moving_variance", l.mv)):
                    self.params[var_name(l.idx, "BatchNorm/" + leaf)] = t
                l.scale = torch.empty(l.cout, dtype=F32, device=dev)
                l.shift = torch.empty(l.cout, dtype=F32, device=dev)
                l.mean = torch.zeros(l.cout, dtype=F32, device=dev)
                l.rstd = torch.ones(l.cout, dtype=F32, device=dev)
        assert off_d == n_decay and off_n == self.n_params

    def trainable_names(self) -> List[str]:
        return list(self.arena_slices)

    @property
    def learning_rate(self) -> float:
        """AdamOptimizer(learning_rate) (train_yolo3_mask.py:38,55).  The optimizer kernel reads it from
        device memory, so it can be changed between steps of a recorded program -- which the reference's
        own schedule (train_yolo3_mask.py:130-141) never achieves: its graph captured the initial 1e-4
        (SURVEY F6)."""
        return self._lr

    @learning_rate.setter
    def learning_rate(self, value: float) -> None:
        self._lr = float(value)
        if self.lr_dev is not None:
            self.sync_lanes()        # (an open tail's Adam sweeps still read lr_dev on the side lane)
            self.lr_dev.fill_(self._lr)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Variables under the reference's checkpoint names/shapes (weights, BN gamma/beta/
        moving stats, biases; no optimizer slots -- train_yolo3_mask.py:41-58)."""
        self.sync_lanes()
        return {k: v.detach().clone() for k, v in self.params.items()}

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        self.sync_lanes()
        for k, v in sd.items():
            if k not in self.params:
                if strict:
                    raise KeyError(k)
                continue
            self.params[k].copy_(torch.as_tensor(v).to(self.device, F32).reshape(self.params[k].shape))
        if strict:
            missing = set(self.params) - set(sd)
            if missing:
                raise KeyError("missing variables: %s" % sorted(missing)[:4])
        if not self.plan_only:
            self.refresh_weights()

    # ------------------------------------------------------------------ static plan
    def _plan(self, B: int, S: int) -> None:
        """Allocate every activation / gradient buffer and build the conv descriptors once
        for (B, S): nothing is allocated inside forward / train_step."""
        dev = self.device
        self.B, self.S = B, S
        self.images = torch.zeros(2 * B if self.pair else B, S, S, 3, dtype=F32, device=dev)
        spatial = {0: (S, S)}
        # which activations need a gradient: any trainable layer at or upstream of a consumer
        for l in self.layers:
            H, W = spatial[l.src]
            l.H, l.W = H, W
            l.Ho, l.pad_t = L.same_pads(H, l.k, l.stride)
            l.Wo, l.pad_l = L.same_pads(W, l.k, l.stride)
            spatial[l.idx] = (l.Ho, l.Wo)
        has_trainable_upto = {0: False}
        for l in self.layers:
            up = has_trainable_upto[l.src]
            if l.src_up is not None:
                up = up or has_trainable_upto[l.src_up]
            if l.shortcut is not None:
                up = up or has_trainable_upto[l.shortcut]
            has_trainable_upto[l.idx] = up or (not l.lock)
        self._needs_grad_into = {i: v for i, v in has_trainable_upto.items()}
        if self.training:
            for l in self.layers:
                ups = [l.src] + ([l.src_up] if l.src_up is not None else [])
                if l.lock and any(has_trainable_upto[u] for u in ups):
                    raise NotImplementedError(
                        "layer %d is locked but has trainable layers upstream; only the reference's two stages "
                        "(conv1-52 locked, or nothing locked) are supported (yolo/yolo3_net_pos.py:155-156)" % l.idx)
        # activation tensors are views into ONE allocation: a process that has allocated and freed a lot of
        # device memory gets later allocations on worse-mapped memory (B = 32 inference 4.8 -> 3.9 k img/s after
        # 13 GB of tensors were created and freed, no compute involved: tools/micro/infer_after_alloc.py), and
        # one large region fares better than eighty medium ones
        use_arena = os.environ.get("DISYOLO_ARENA", "1") != "0"
        need = 0
        batch_of = lambda l: 2 * B if (self.pair and l.idx <= self._pair_P) else B
        for l in self.layers:
            n = batch_of(l) * l.Ho * l.Wo * l.cout
            tb = self.training and (not l.lock) and l.kind != "lin"
            need += ((n * (4 if l.kind == "lin" else 2) + 255) // 256) * 256 * (2 if tb else 1)
        self._act_arena = torch.zeros(need, dtype=torch.uint8, device=dev) if use_arena else None
        cursor = [0]

        def zeros4(b, h, w, c, dtype):
            if self._act_arena is None:
                return torch.zeros(b, h, w, c, dtype=dtype, device=dev)
            nbytes = b * h * w * c * (4 if dtype == F32 else 2)
            t = self._act_arena[cursor[0]:cursor[0] + nbytes].view(dtype).view(b, h, w, c)
            cursor[0] += ((nbytes + 255) // 256) * 256
            return t

        for l in self.layers:
            M = B * l.Ho * l.Wo
            train_bn = self.training and (not l.lock) and l.kind != "lin"
            if l.kind == "lin":
                l.act = zeros4(B, l.Ho, l.Wo, l.cout, F32)
            else:
                l.act = zeros4(batch_of(l), l.Ho, l.Wo, l.cout, BF16)
                l.raw = zeros4(B, l.Ho, l.Wo, l.cout, BF16) if train_bn else None
            if l.idx > 1:
                K = l.k * l.k * l.cin
                l.wp = torch.zeros(l.cout, K, dtype=BF16, device=dev)
            l.cout_pad = l.cout if l.cout % 32 == 0 else ((l.cout + 31) // 32) * 32
            if self.training and not l.lock:
                # gradient wrt this layer's conv output (bf16, row pitch cout_pad)
                if l.kind == "lin":
                    l.dx = torch.zeros(B, l.Ho, l.Wo, L.GRAD_LD, dtype=BF16, device=dev)
                else:
                    l.dx = torch.zeros(B, l.Ho, l.Wo, l.cout, dtype=BF16, device=dev)
            if self.training and has_trainable_upto[l.idx] and l.kind != "lin":
                l.grad = torch.zeros(B, l.Ho, l.Wo, l.cout, dtype=BF16, device=dev)
            needs_dgrad = self.training and (not l.lock) and l.idx > 1 and (
                has_trainable_upto[l.src] or (l.src_up is not None and has_trainable_upto[l.src_up]))
            if needs_dgrad:
                l.wdg = torch.zeros(l.cin, l.k * l.k * l.cout_pad, dtype=BF16, device=dev)
        self._build_descs()
        if self.dtype == "fp8":
            self._plan_fp8()
        l1 = self.by_idx[1]
        if self.training and not l1.lock:
            self._img8 = torch.zeros(B, S, S, 8, dtype=BF16, device=dev)
            self._dw8 = torch.zeros(3, 3, 8, l1.cout, dtype=F32, device=dev)
            self._wgrad1_desc = L.make_conv_desc(self._img8, l1.dx, l1.dx, 3, 1)
            l1.stats_rows = L.colstats_rows(B * S * S, l1.cout)
            l1.stats = torch.zeros(l1.stats_rows, l1.cout, 2, dtype=F32, device=dev)
            self._ones32 = torch.ones(l1.cout, dtype=F32, device=dev)
            self._zeros32 = torch.zeros(l1.cout, dtype=F32, device=dev)
        # detection / loss buffers
        self.clip_window = torch.zeros(B, 4, dtype=F32, device=dev)
        self.detections = torch.zeros(B, cfg.MAX_DETECTION, 6, dtype=F32, device=dev)
        self.det_count = torch.zeros(B, dtype=torch.int32, device=dev)
        self.masks = None
        self.keep = torch.zeros(B, cfg.MAX_DETECTION, dtype=torch.int32, device=dev)
        if self.training:
            G = cfg.MAX_BOX_PER_IMAGE
            g1 = S // 32
            self.labels = [torch.zeros(B, g, g, 3, 5 + self.num_class, dtype=F32, device=dev)
                           for g in (4 * g1, 2 * g1, g1)]          # yolo3, yolo2, yolo1
            self.true_boxes = torch.zeros(B, G, 5, dtype=F32, device=dev)
            self.true_masks = torch.zeros(B, G, S, S, dtype=torch.uint8, device=dev)
            self.perm_det = torch.arange(cfg.MAX_DETECTION, dtype=torch.int32, device=dev).repeat(B, 1).contiguous()
            self.perm_gt = torch.arange(G, dtype=torch.int32, device=dev).repeat(B, 1).contiguous()
            self.rois = torch.zeros(B, L.ROI_MAX, L.ROI_W, dtype=torch.int32, device=dev)
            self.roi_count = torch.zeros(B, dtype=torch.int32, device=dev)
            self._in_sets = None
            if self.pair:     # the per-batch inputs of the trainable part, once per half
                names = ("labels", "true_boxes", "true_masks", "perm_det", "perm_gt", "clip_window")
                first = {n: getattr(self, n) for n in names}
                second = {n: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for n, v in first.items()}
                self._in_sets = [first, second]
            self.losses = torch.zeros(8, dtype=F32, device=dev)
            self.mask_loss = torch.zeros(1, dtype=F32, device=dev)
            self.reg_loss = torch.zeros(1, dtype=F32, device=dev)
            # total loss of the last LOSS_RING steps, filed by the optimizer's finish (step_losses())
            self.loss_ring = torch.zeros(self.LOSS_RING, dtype=F32, device=dev)
            # temp for the data gradient of the fused upsample+concat layers (full-res, C1 ch)
            tmax = 0
            for l in self.layers:
                if l.src_up is not None and l.wdg is not None:
                    tmax = max(tmax, B * l.H * l.W * self.by_idx[l.src_up].cout)
            self.up_tmp = torch.zeros(max(tmax, 1), dtype=BF16, device=dev)
            # size the shared workspace once (wgrad slabs dominate)
            need = 1 << 20
            for l in self.layers:
                if l.wgrad_desc is not None:
                    need = max(need, L.conv2d_wgrad_workspace(l.wgrad_desc))
                if l.dx is not None and l.kind != "lin":
                    need = max(need, L.load().disyolo_bn_act_bwd_workspace(B * l.Ho * l.Wo, l.cout))
            if not l1.lock:
                need = max(need, L.conv2d_wgrad_workspace(self._wgrad1_desc),
                           L.load().disyolo_conv_first_wgrad_workspace(B, S, S, l1.cout))
            self.ws.get(int(need))
            self.ws_aux.get(int(need))
        self.ws_det.get(int(max(L.load().disyolo_detect_workspace(B, S, self.num_class), 1 << 20)))
        self._build_dgrad_descs()
        self.refresh_weights()

    # ---- fp8 path of the inference-mode backbone --------------------------------------------
    # Round 6: conv1-9 stay bf16 -- their fused launches (conv1+2, the 288^2 / 144^2 residual blocks) beat any per-layer fp8
    # kernel, and their inputs have 3 ... 64 channels, below the 128-byte K slice of the block-scaled MFMA -- and conv10-52 run
    # e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4 (csrc/conv_fp8.hip, MX): conv9's bf16 output is quantised once for conv10.
    # DISYOLO_FP8_FROM=1 gives round 5's all-fp8 backbone (profiles/r06_fp8_mx_layers.txt has both).
    FP8_FROM = int(os.environ.get("DISYOLO_FP8_FROM", "10"))
    FP8_UPTO = 52
    FP8_DUAL = (4, 9, 26, 43, 52)     # outputs that bf16 layers consume too (skip2..5, the trunk's end)

    def _fp8_layers(self):
        """layers that run in fp8: conv FP8_FROM..52 when they are in inference mode (locked, or an inference net)"""
        return [l for l in self.layers if self.FP8_FROM <= l.idx <= self.FP8_UPTO and (l.lock or not self.training)]

    def _plan_fp8(self) -> None:
        dev = self.device
        ls = self._fp8_layers()
        if [l.idx for l in ls] != list(range(self.FP8_FROM, self.FP8_UPTO + 1)) or not all(
                l.lock or not self.training for l in self.layers[:self.FP8_UPTO]):
            raise L.DisyoloError("dtype='fp8' needs conv1-52 in inference mode (stage 1 training, or training=False)")
        # the bf16 layer in front of the first fp8 layer: its output is also kept as e4m3 (one quantisation pass per forward)
        self._fp8_entry = self.by_idx[self.FP8_FROM - 1] if self.FP8_FROM > 1 else None
        if self._fp8_entry is not None:
            q = self._fp8_entry
            q.act8 = torch.zeros(self.B, q.Ho, q.Wo, q.cout, dtype=torch.uint8, device=dev)
            q.s_out = 1.0
        for l in ls:
            l.act8 = torch.zeros(self.B, l.Ho, l.Wo, l.cout, dtype=torch.uint8, device=dev)
            l.escale = torch.zeros(l.cout, dtype=F32, device=dev)
            l.dual16 = l.idx in self.FP8_DUAL
            l.s_out, l.s_w = 1.0, 1.0
            if l.idx > 1:
                l.w8 = torch.zeros(l.cout, l.k * l.k * l.cin, dtype=torch.uint8, device=dev)
                l.desc8 = L.make_conv_desc(self.by_idx[l.src].act8, l.w8, l.act8, l.k, l.stride, leaky=True, alpha=cfg.ALPHA)

    def calibrate_fp8(self, margin: float = 1.0) -> Dict[int, float]:
        """Per-tensor scales from the batch currently set: runs conv1-52 in bf16 once, takes max|activation|
        of every layer (after the residual add) and max|weight|; scale = max / (448 * margin).  Static
        afterwards (the step can be recorded).  Returns {layer: activation scale}."""
        if self.dtype != "fp8":
            raise L.DisyoloError("calibrate_fp8 on a bf16 net")
        self.sync_lanes()
        self.fp8_ready = False
        for l in self.layers[:self.FP8_UPTO]:       # (layer by layer in bf16: every output is written)
            self._forward_layer(l, False)
        torch.cuda.synchronize()
        scaled = self._fp8_layers() + ([self._fp8_entry] if self._fp8_entry is not None else [])
        for l in scaled:
            amax = float(l.act.float().abs().max())
            l.s_out = max(amax, 1e-12) / (448.0 * margin)
        for l in self._fp8_layers():
            if l.idx > 1:
                l.s_w = max(float(l.w.abs().max()), 1e-12) / 448.0
        self.fp8_ready = True
        self.refresh_weights()
        return {l.idx: l.s_out for l in sorted(scaled, key=lambda x: x.idx)}

    def _forward_layer_fp8(self, l) -> None:
        if l.idx == 1:
            L.conv_first_fwd_fp8(self.images, l.w, l.scale, l.shift, l.act8, l.s_out, alpha=cfg.ALPHA)
            return
        sc = self.by_idx[l.shortcut] if l.shortcut is not None else None
        L.conv2d_fp8_fwd(l.desc8, l.w8, l.escale, l.shift, l.act8, l.s_out, y16=l.act if l.dual16 else None,
                         residual8=sc.act8 if sc is not None else None, residual_scale=sc.s_out if sc is not None else 0.0)

    def _build_descs(self) -> None:
        """conv descriptors (raw pointers into the activation buffers).  Rebuilt when the
        pipelined step switches the double-buffered backbone outputs (``_use_parity``)."""
        dev = self.device
        for l in self.layers:
            if l.idx == 1:
                continue
            x0 = self._input_of(l, l.src)
            x1 = self._input_of(l, l.src_up) if l.src_up is not None else None
            res = self._input_of(l, l.shortcut) if l.shortcut is not None else None
            train_bn = self.training and (not l.lock) and l.kind != "lin"
            if l.kind == "lin":
                l.desc = L.make_conv_desc(x0, l.wp, l.act, l.k, l.stride, x1=x1, shift=l.bias, out_f32=True)
            elif train_bn:
                if l.stats is None:
                    d0 = L.make_conv_desc(x0, l.wp, l.raw, l.k, l.stride, x1=x1)
                    l.stats_rows = L.conv2d_stats_rows(d0)
                    l.stats = torch.zeros(l.stats_rows, l.cout, 2, dtype=F32, device=dev)
                l.desc = L.make_conv_desc(x0, l.wp, l.raw, l.k, l.stride, x1=x1, stats=l.stats)
                l.fused_fwd = False
                if (self.bn_inkernel and not self.sync_bn and res is None and l.idx not in self.HEAD_LAYERS
                        and not getattr(self, "plan_only", False) and l.stride == 1):
                    # the tile: the table's if its grid can run the in-launch exchange (one block per CU, all resident),
                    # else the covering tile with the largest such grid (the statistics buffer follows the tile's rows)
                    tile = self._fused_tile(lambda t: L.make_conv_desc(x0, l.wp, l.raw, l.k, l.stride, x1=x1, tile=t), l.k)
                    if tile is not None:
                        d0 = L.make_conv_desc(x0, l.wp, l.raw, l.k, l.stride, x1=x1, tile=tile)
                        rows = L.conv2d_stats_rows(d0)
                        l.stats_rows = rows
                        if l.stats.shape[0] < rows:      # (never shrunk: autotune() sizes it for every candidate's rows)
                            l.stats = torch.zeros(rows, l.cout, 2, dtype=F32, device=dev)
                        if l.csync is None:
                            l.csync = L.cluster_sync_buffer(l.cout, dev)
                        l.desc = L.make_conv_desc(x0, l.wp, l.raw, l.k, l.stride, x1=x1, stats=l.stats, tile=tile,
                                                  bn_fused=dict(y_act=l.act, gamma=l.gamma, beta=l.beta, mm=l.mm, mv=l.mv,
                                                                scale=l.scale, shift=l.shift, mean=l.mean, rstd=l.rstd,
                                                                decay=cfg.BN_DECAY, eps=cfg.BN_EPSILON, sync=l.csync),
                                                  alpha=cfg.ALPHA)
                        l.fused_fwd = True
            else:
                l.desc = L.make_conv_desc(x0, l.wp, l.act, l.k, l.stride, x1=x1, scale=l.scale, shift=l.shift,
                                          residual=res, leaky=True, alpha=cfg.ALPHA)
            if self.training and not l.lock:
                l.wgrad_desc = L.make_conv_desc(x0, l.wp, l.act, l.k, l.stride, x1=x1)

    # tiles tried for a launch that runs batch norm in its epilogue when the table's own cannot (its grid must be one block
    # per CU at most): GEMM tiles 64x128, 128x64, 64x64, 96x128, 128x128, 192x128; for 3x3 layers the patch kernels first
    FUSED_TILES_1x1 = (3, 2, 6, 10, 1, 12, 0x20c)
    FUSED_TILES_3x3 = (16, 18, 19, 12, 0x20c)

    def _fused_tile(self, make_desc, k: int, bwd: bool = False):
        """tile code for a conv that carries the in-launch batch norm (forward form, or backward with ``bwd``), or None.
        ``make_desc(tile)`` builds the plain descriptor (tile 0 = the table's / launcher's pick)."""
        best, best_blocks = None, -1
        for i, t in enumerate((0,) + (self.FUSED_TILES_3x3 if k == 3 else self.FUSED_TILES_1x1)):
            d = make_desc(t)
            tid, bm, bn, _, _ = L.conv2d_tile(d)
            if t and tid != (t & 0xff):
                continue                      # (the candidate does not cover the shape: the launcher fell back)
            if bwd:
                d.flags |= L.CONV_BN_BWD_FUSED
            ok = L.conv2d_bn_fused_ok(d)
            d.flags &= ~L.CONV_BN_BWD_FUSED
            if not ok:
                continue
            if i == 0:
                return d.tile                 # the table's own tile runs it: keep it
            blocks = L.conv2d_stats_rows(d) * (-(-d.Cout // bn))
            if blocks > best_blocks:
                best, best_blocks = t, blocks
        return best

    def _input_of(self, l, src: int) -> torch.Tensor:
        """activation of layer ``src`` as layer ``l`` reads it (backbone_pair: a trainable layer sees the current half of
        a backbone output)"""
        a = self.by_idx[src].act
        if self.pair and src <= self._pair_P < l.idx:
            return a[self._half * self.B:(self._half + 1) * self.B]
        return a

    # ---- cross-step software pipeline of the locked backbone (stage 1) -------------------
    def _backbone_prefix(self) -> int:
        """number of leading locked layers whose outputs do not depend on trained weights"""
        p = 0
        for l in self.layers:
            if not l.lock:
                break
            p = l.idx
        return p

    PIPE_INPUTS = ("images", "clip_window", "labels", "true_boxes", "true_masks", "perm_det", "perm_gt")
    feed_stream = None
    _pipe_in = None

    def _setup_pipeline(self) -> None:
        P = self._backbone_prefix()
        if P < 2 or not self.training:
            raise L.DisyoloError("the backbone pipeline needs a locked layer prefix (stage 1)")
        self._pipe_P = P
        # backbone outputs consumed by the trainable part: two copies each
        xs = sorted({i for l in self.layers if l.idx > P for i in (l.src, l.src_up, l.shortcut)
                     if i is not None and 1 <= i <= P})
        self._xbuf = {i: (self.by_idx[i].act, torch.zeros_like(self.by_idx[i].act)) for i in xs}
        self._parity = 0
        # the per-batch inputs, two sets: the list of parity q reads set q (labels / boxes / masks of its batch, images of the
        # batch after it), so the feed of the NEXT replay can be written while this one runs (``feed_context``).  Set 1 starts
        # as a copy of set 0: a batch set once before build_program is replayed by both lists.
        first = {n: getattr(self, n) for n in self.PIPE_INPUTS}
        second = {n: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for n, v in first.items()}
        self._pipe_in = [first, second]
        # events (created once, re-recorded): [q] the last reader of set q is done (recorded behind the replay / the priming
        # pass); set q was written on the feed stream (the next replay of parity q waits for it)
        cuda = self.device.type == "cuda"
        self._ev_free = [torch.cuda.Event() for _ in (0, 1)] if cuda else None
        self._ev_fed = [torch.cuda.Event() for _ in (0, 1)] if cuda else None
        self._free_valid = [False, False]
        self._fed_pending = [False, False]
        self._run_stream = None
        self.feed_stream = None            # (chosen once the lists exist: _pick_feed_stream)

    def _pick_feed_stream(self) -> None:
        """the stream ``feed_context()`` puts the next batch on.  Single-GPU: lane 3 of the executor's pool (the gradient exchange's
        lane, idle without data parallelism) -- a stream the pool PROBED to run beside the main lane and not to slow it while
        parked on an event (csrc/runtime.hip pool_lane; profiles/r05_hw_queues.txt); an arbitrary new stream can be the other
        kind of neighbour.  (Measured the same as a plain stream for the feeds of this round, tools/feed_rate.py; what did cost a
        step 7.1 ms instead of 3.6 was a SECOND stream parked on a 2-ms upload's event -- the host feeder now uploads on this
        stream too.)  Data parallel: lane 3 is taken, a plain lowest-priority stream has to do."""
        if self.device.type != "cuda":
            return
        if self.dp is None and os.environ.get("DISYOLO_FEED_LANE", "1") != "0":
            self.feed_stream = self._progs[0][0].lane_stream(L.COMM_LANE, self.device)
            return
        try:
            prio = max(torch.cuda.Stream.priority_range())       # (largest number = lowest priority)
        except Exception:
            prio = 0
        self.feed_stream = torch.cuda.Stream(device=self.device, priority=prio)

    def _use_parity(self, q: int) -> None:
        for i, bufs in self._xbuf.items():
            self.by_idx[i].act = bufs[q]
        if self._pipe_in is not None:
            for n, v in self._pipe_in[q].items():
                setattr(self, n, v)
        self._build_descs()

    def prime_pipeline(self, images=None, clip_window=None) -> None:
        """run the backbone once for ``images`` (default: the images last set) -- fills parity 0; afterwards every
        train_step consumes that result and computes the backbone of the NEXT images meanwhile"""
        src = self.images
        self._use_parity(0)
        if images is not None:
            self._set_inputs(images, clip_window if clip_window is not None else self.clip_window)
        elif src is not self.images:
            self.images.copy_(src)
        self._forward_prefix(self._pipe_P, True)
        self._parity = 0
        if self.device.type == "cuda":
            for q in (0, 1):
                # (both sets: whatever the caller's stream did before -- a validation sweep, an eager step -- is in front of it)
                self._ev_free[q].record(torch.cuda.current_stream())
                self._free_valid[q] = True
                self._fed_pending[q] = False
            self._run_stream = torch.cuda.current_stream()

    def feed_context(self):
        """``with net.feed_context(): batch = data.get(); net.set_batch(batch)`` -- in the pipelined step the batch of the NEXT
        replay is produced and copied on a stream of its own, beside the replay that is running (two input sets, one per list);
        ``train_step(None)`` then waits for that stream on the device.  Anywhere else: no-op, set_batch runs on the caller's
        stream in front of the step, as before."""
        if (self._progs is not None and self._pipe_in is not None and self.feed_stream is not None
                and os.environ.get("DISYOLO_FEED_STREAM", "1") != "0"):
            return torch.cuda.stream(self.feed_stream)
        return contextlib.nullcontext()

    def _build_dgrad_descs(self) -> None:
        """Data-gradient convs: forward kernel over dx with the flipped operand (wdg), pads
        k-1-pad and the transposed gather (in_div = stride)."""
        for l in self.layers:
            l.dgrad_descs = []
            if l.wdg is None:
                continue
            Kp = l.k * l.k * l.cout_pad
            pads = (l.k - 1 - l.pad_t, l.k - 1 - l.pad_l)
            if l.src_up is None:
                tgt = self.by_idx[l.src]
                kw = dict(w=l.wdg, pads=pads, out_hw=(l.H, l.W))
                # the shallow stride-2 layers (conv2, conv5): data gradient as one 2x2-tap conv over dy (lib.dgrad_s2_quad)
                if (l.k == 3 and l.stride == 2 and l.H % 2 == 0 and l.W % 2 == 0 and l.pad_t == 0 and l.pad_l == 0
                        and os.environ.get("DISYOLO_DGRAD_QUAD", "1") != "0"
                        and L.dgrad_s2_quad_ok(self.B, l.Ho, l.Wo, l.cout, l.cin)):
                    if getattr(l, "wq", None) is None:
                        l.wq = torch.zeros(4 * l.cin, 9 * l.cout, dtype=BF16, device=self.device)
                    kw["quad"] = l
                l.dgrad_descs.append(("direct", tgt, kw))
            else:
                skip, up = self.by_idx[l.src], self.by_idx[l.src_up]
                if self._needs_grad_into[skip.idx]:
                    l.dgrad_descs.append(("direct", skip, dict(w=l.wdg[:skip.cout], pads=pads, out_hw=(l.H, l.W))))
                if self._needs_grad_into[up.idx]:
                    tmp = self.up_tmp[:self.B * l.H * l.W * up.cout].view(self.B, l.H, l.W, up.cout)
                    l.dgrad_descs.append(("up", up, dict(w=l.wdg[skip.cout:skip.cout + up.cout], pads=pads,
                                                         out_hw=(l.H, l.W), tmp=tmp)))

    def refresh_weights(self) -> None:
        """(Re)pack the bf16 MFMA operands from the f32 masters and fold the locked /
        inference batch-norm statistics into per-channel scale/shift."""
        self.sync_lanes()
        for l in self.layers:
            if l.idx > 1:
                L.pack_weights(l.w, l.wp, l.wdg, l.k, l.cin, l.cout, l.cout_pad)
            if l.kind != "lin":
                L.bn_fold(l.gamma, l.beta, l.mm, l.mv, cfg.BN_EPSILON, l.scale, l.shift)
        if self.dtype == "fp8" and self.fp8_ready:
            for l in self._fp8_layers():
                if l.idx > 1:
                    L.pack_weights_fp8(l.w, l.w8, l.k, l.cin, l.cout, l.s_w)
                    # acc is in units of s_in * s_w: fold them into the batch-norm scale
                    torch.mul(l.scale, self.by_idx[l.src].s_out * l.s_w, out=l.escale)

    # ------------------------------------------------------------------ forward
    # layers that only feed a detection head: in a recorded step they run on the side lane while
    # the trunk continues (58/59 after 57, 66/67 after 65, 74/75 after 73)
    HEAD_BRANCH = {58: 57, 66: 65, 74: 73}
    HEAD_LAYERS = (58, 59, 66, 67, 74, 75)

    def _forward_layers(self, is_training: bool, first: int = 1) -> None:
        plan = self._fusion_plan(is_training, first, 82)
        xwaits = []
        for l in self.layers:
            if l.idx < first:
                continue
            if self.use_side_lane:
                if l.idx in self.HEAD_BRANCH:
                    L.lane_sync(0, 1)        # the branch point's output is ready on the main lane
                    L.set_lane(1)
                elif l.idx in (60, 68, 76):
                    L.set_lane(0)
            if self._overlap_rec:
                # cross-replay waits are emitted in front of the first launch that EXECUTES at or after their layer (a layer
                # fused away into a later launch -- plan entry None -- carries its wait forward, it is never dropped)
                if l.idx in self._xstep_waits:
                    xwaits.append(self._xstep_waits[l.idx])         # the previous replay's last reader of this layer's output
                if l.idx == self._xstep_first_trainable and not self._xstep_all_merged:
                    xwaits.append(self.SLOT_ALL)                     # the previous replay's optimizer + re-pack
                if xwaits and (l.idx not in plan or plan[l.idx] is not None):
                    for slot in xwaits:
                        L.lane_wait_slot(slot, 0)
                    xwaits.clear()
            if l.idx in plan:
                if plan[l.idx] is not None:
                    plan[l.idx]()
                self._fp8_handover(l)
            else:
                self._forward_layer(l, is_training)
                self._fp8_handover(l)
            hook = getattr(self, "_fwd_hook", None)
            if hook is not None and hook[0] == l.idx and L.CURRENT_LANE == 0:
                self._fwd_hook = None
                hook[1]()
        if self.use_side_lane:
            L.set_lane(0)
            L.lane_sync(1, 0)

    def _fold_trainable(self, *idxs) -> None:
        """a fused launch reads scale / shift of its layers directly: in a training net evaluated with is_training=False a
        trainable layer's pair still holds the last step's BATCH statistics -- fold its moving statistics first (what
        _forward_layer does for the layer-by-layer path)"""
        for i in idxs:
            l = self.by_idx[i]
            if self.training and not l.lock and l.kind != "lin":
                L.bn_fold(l.gamma, l.beta, l.mm, l.mv, cfg.BN_EPSILON, l.scale, l.shift)

    SLOT_ALL = 15

    def _plan_overlap(self) -> None:
        """cross-replay dependencies of the overlapped tail: backbone outputs that trainable layers consume (their weight
        gradients read them on the side lane) -> slot; the first trainable layer"""
        P = self._backbone_prefix()
        self._xstep_first_trainable = P + 1
        self._xstep_slot, self._xstep_reader = {}, {}
        slot = 0
        for l in self.layers:
            if l.idx <= P:
                continue
            for src in (l.src, l.src_up, l.shortcut):
                if src is not None and 1 <= src <= P:
                    if src not in self._xstep_slot:
                        self._xstep_slot[src] = slot
                        slot += 1
                    # the LAST reader in backward order marks the slot (side lane, FIFO: earlier readers are done by then)
                    self._xstep_reader[src] = l.idx
        if slot >= self.SLOT_ALL:
            raise L.DisyoloError("overlap_tail: too many backbone outputs feed trainable layers")
        order = [x.idx for x in self.backward_order() if not x.lock]
        for src in self._xstep_slot:
            readers = [x.idx for x in self.layers if x.idx > P and src in (x.src, x.src_up, x.shortcut)]
            self._xstep_reader[src] = max(readers, key=order.index)
        # where the next replay's main lane waits.  A wait that is enqueued before its event has completed costs the waiting
        # stream ~3 us whether or not it ever blocks (tools/micro/event_cost.hip), so the waits are merged: the side lane is
        # FIFO, hence ONE wait -- in front of the first of these backbone layers -- on the slot whose reader comes latest in
        # the backward pass covers all of them; a source whose last reader is the backward pass's LAST layer can only be
        # released by the end of the tail: the wait for the whole tail moves up to it (one backbone layer less of overlap)
        self._xstep_waits = {src: s for src, s in self._xstep_slot.items()}
        self._xstep_all_merged = False
        if self.use_side_lane and self.tail_on_main == 0 and self._xstep_slot and os.environ.get("DISYOLO_XSTEP_MERGE", "1") != "0":
            late = [src for src in self._xstep_slot if self._xstep_reader[src] == order[-1]]
            early = [src for src in self._xstep_slot if src not in late]
            self._xstep_waits = {}
            if early:
                last = max(early, key=lambda src: order.index(self._xstep_reader[src]))
                self._xstep_waits[min(early)] = self._xstep_slot[last]
            if late:
                self._xstep_waits[min(late)] = self.SLOT_ALL
                self._xstep_all_merged = True

    def _forward_first_two(self) -> None:
        l1, l2 = self.by_idx[1], self.by_idx[2]
        self._fold_trainable(1, 2)
        L.conv12_fused_fwd(self.images, l1.w, l1.scale, l1.shift, l2.wp, l2.scale, l2.shift, l2.act, alpha=cfg.ALPHA)

    def _forward_block34(self) -> None:
        l2, l3, l4 = self.by_idx[2], self.by_idx[3], self.by_idx[4]
        self._fold_trainable(3, 4)
        L.block32_fused_fwd(self._input_of(l3, 2), None, l3.wp, l3.scale, l3.shift, l4.wp, l4.scale, l4.shift, l4.act, post=0, alpha=cfg.ALPHA)

    def _forward_block64(self, i3: int) -> None:
        la, lb = self.by_idx[i3 - 1], self.by_idx[i3]
        self._fold_trainable(i3 - 1, i3)
        L.block64_fused_fwd(self._input_of(la, la.src), la.wp, la.scale, la.shift, lb.wp, lb.scale, lb.shift, lb.act, alpha=cfg.ALPHA)

    def _forward_mask_head(self) -> None:
        l80, l81, l82 = self.by_idx[80], self.by_idx[81], self.by_idx[82]
        self._fold_trainable(80, 81)
        L.block32_fused_fwd(self._input_of(l80, l80.src), self._input_of(l80, l80.src_up), l80.wp, l80.scale, l80.shift, l81.wp, l81.scale,
                            l81.shift, l82.act, post=1, wC=l82.wp, biasC=l82.bias, alpha=cfg.ALPHA)

    def _forward_prefix(self, upto: int, is_training: bool) -> None:
        """layers 1..upto one after the other on the current lane (the pipelined backbone), same kernels as
        _forward_layers"""
        plan = self._fusion_plan(is_training, 1, upto)
        for l in self.layers[:upto]:
            if l.idx in plan:
                if plan[l.idx] is not None:
                    plan[l.idx]()
                self._fp8_handover(l)
                continue
            self._forward_layer(l, is_training)
            self._fp8_handover(l)

    def _fp8_handover(self, l) -> None:
        """the bf16 layer in front of the first fp8 layer: its output once more as e4m3 (what conv FP8_FROM reads)"""
        if self.dtype == "fp8" and self.fp8_ready:
            q = getattr(self, "_fp8_entry", None)
            if q is not None and l.idx == q.idx:
                L.quant_fp8(q.act, q.act8, q.s_out)

    def _inference_mode(self, idxs, is_training: bool) -> bool:
        """every one of these layers normalises with its folded moving statistics in this pass"""
        return all(self.by_idx[i].kind == "lin" or not (is_training and self.training and not self.by_idx[i].lock) for i in idxs)

    def _fusion_plan(self, is_training: bool, first: int, last: int):
        """layer index -> fused launch that replaces the layer (None: covered by a later entry).  Groups of layers whose
        intermediates have a single consumer and whose batch norms are in inference mode in this pass: conv1+conv2, the
        first residual block (conv3+conv4), the mask head (conv80+81+82) -- bf16 path only."""
        plan = {}
        if self.dtype == "fp8" and self.fp8_ready:
            # the fp8 layers run one by one; the bf16 layers in front of them (conv1-9) keep their fused launches
            bf16_plan = self._fusion_plan_bf16(is_training, first, last)
            groups, cur = [], []
            for i in sorted(bf16_plan):
                cur.append(i)
                if bf16_plan[i] is not None:
                    groups.append(cur)
                    cur = []
            for g in groups:
                if not any(self.FP8_FROM <= i <= self.FP8_UPTO for i in g):
                    for i in g:
                        plan[i] = bf16_plan[i]
            return plan
        return self._fusion_plan_bf16(is_training, first, last)

    def _fusion_plan_bf16(self, is_training: bool, first: int, last: int):
        plan = {}
        if first <= 1 and last >= 2 and self._can_fuse_first_two(is_training):
            plan[1], plan[2] = None, self._forward_first_two
        if self.fuse_blocks and not getattr(self, "plan_only", False):
            l2, l4 = self.by_idx[2], self.by_idx[4]
            if (first <= 3 and last >= 4 and self._inference_mode((3, 4), is_training)
                    and L.block32_fused_ok(self.B, l4.Ho, l4.Wo, l2.cout, 0, 0)):
                plan[3], plan[4] = None, self._forward_block34
            for i3 in ((7, 9) if os.environ.get("DISYOLO_FUSE_B64", "1") != "0" else ()):   # the 144^2 residual blocks: conv6+7, conv8+9
                lb = self.by_idx[i3]
                if (first <= i3 - 1 and last >= i3 and self._inference_mode((i3 - 1, i3), is_training)
                        and L.block64_fused_ok(self.B, lb.Ho, lb.Wo, self.by_idx[i3 - 2].cout)):
                    plan[i3 - 1], plan[i3] = None, (lambda i3=i3: self._forward_block64(i3))
            l80, l82 = self.by_idx[80], self.by_idx[82]
            if (first <= 80 and last >= 82 and self._inference_mode((80, 81, 82), is_training) and l82.cout == 9
                    and L.block32_fused_ok(self.B, l82.Ho, l82.Wo, self.by_idx[l80.src].cout, self.by_idx[l80.src_up].cout, 1)):
                plan[80], plan[81], plan[82] = None, None, self._forward_mask_head
        return plan

    def _can_fuse_first_two(self, is_training: bool) -> bool:
        """conv1 and conv2 both in inference mode (folded moving statistics), bf16 path, a size the fused kernel covers"""
        if not self.fuse_first_two or (self.dtype == "fp8" and self.fp8_ready and self.FP8_FROM <= 2):
            return False
        if not self._inference_mode((1, 2), is_training):
            return False            # (a trainable conv1 / conv2 needs its own output -- and its batch statistics)
        return L.conv12_fused_ok(self.B, self.S, self.S)

    def _forward_layer(self, l, is_training: bool) -> None:
        B = self.B
        if self.dtype == "fp8" and self.fp8_ready and l.act8 is not None and l.idx >= self.FP8_FROM:
            self._forward_layer_fp8(l)
            return
        train_bn = is_training and self.training and (not l.lock) and l.kind != "lin"
        M = B * l.Ho * l.Wo
        res = self._input_of(l, l.shortcut) if l.shortcut is not None else None
        if l.idx == 1:
            if train_bn:
                L.conv_first_fwd(self.images, l.w, self._ones32, self._zeros32, l.raw, alpha=1.0)
                L.colstats(l.raw, l.stats, M, l.cout)
                self._bn_finalize(l, M)
                L.bn_act_fwd(l.raw, l.scale, l.shift, None, l.act, M, l.cout, cfg.ALPHA)
            else:
                if not l.lock and self.training:
                    L.bn_fold(l.gamma, l.beta, l.mm, l.mv, cfg.BN_EPSILON, l.scale, l.shift)
                L.conv_first_fwd(self.images, l.w, l.scale, l.shift, l.act, alpha=cfg.ALPHA)
            return
        if l.kind == "lin":
            L.conv2d_fwd(l.desc)
        elif train_bn:
            L.conv2d_fwd(l.desc)                       # raw conv + per-channel partial sums
            if l.fused_fwd and L.TUNER is None:
                return                                 # ... and, in the same launch, the statistics, the moving averages and the activation
            self._bn_finalize(l, M)
            L.bn_act_fwd(l.raw, l.scale, l.shift, res, l.act, M, l.cout, cfg.ALPHA)
        else:
            if self.training and not l.lock:
                # a training-mode plan evaluated with is_training=False: moving statistics
                L.bn_fold(l.gamma, l.beta, l.mm, l.mv, cfg.BN_EPSILON, l.scale, l.shift)
                d = L.make_conv_desc(self._input_of(l, l.src), l.wp, l.act, l.k, l.stride,
                                     x1=self._input_of(l, l.src_up) if l.src_up is not None else None,
                                     scale=l.scale, shift=l.shift, residual=res, leaky=True, alpha=cfg.ALPHA)
                L.conv2d_fwd(d)
            else:
                L.conv2d_fwd(l.desc)

    def _bn_finalize(self, l, M: int) -> None:
        """batch statistics -> scale/shift/mean/rstd + moving statistics (yolo/yolo3_net_pos.py:90-98).  With
        SyncBN the per-channel sums are added up over the data-parallel ranks first (SURVEY.md 8e option)."""
        rows = l.stats_rows
        if L.TUNER is not None:
            rows = L.TUNER.stats_rows.get(l.stats.data_ptr(), rows)     # rows the candidate tile just wrote
        if not self.sync_bn:
            L.bn_finalize(l.stats, rows, l.cout, M, l.gamma, l.beta, l.mm, l.mv, cfg.BN_DECAY,
                          cfg.BN_EPSILON, l.scale, l.shift, l.mean, l.rstd)
            return
        L.bn_partial_sums(l.stats, rows, l.cout, l.bn_sums)
        self._sync_sums(l.bn_sums)
        L.bn_finalize_sums(l.bn_sums, l.cout, M * self.dp.world_size, l.gamma, l.beta, l.mm, l.mv, cfg.BN_DECAY,
                           cfg.BN_EPSILON, l.scale, l.shift, l.mean, l.rstd)

    def _sync_sums(self, t: torch.Tensor) -> None:
        """all-reduce(SUM) of a small f64 tensor over the ranks, ordered on the lane that produced it: issued
        directly in the per-call path, a cut point of the recorded list otherwise (run_program issues it)"""
        if self.dp.inlist:
            self.dp.comm.allreduce(t)        # a command of the current lane (a launch on the current stream when not recording)
        elif self._rec is not None:
            prog, marks = self._rec
            marks.append((prog.size(), ("sync", t, L.CURRENT_LANE)))
        else:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.dp.pg)

    def enable_sync_bn(self) -> None:
        """called by enable_data_parallel(sync_bn=True)"""
        if self.dp is None:
            raise L.DisyoloError("SyncBN needs data parallelism (enable_data_parallel)")
        if self._prog is not None:
            raise L.DisyoloError("enable SyncBN before build_program()")
        for l in self.layers:
            if self.training and not l.lock and l.kind != "lin":
                l.bn_sums = torch.zeros(l.cout, 2, dtype=torch.float64, device=self.device)
                l.bwd_local = torch.zeros(l.cout, 2, dtype=torch.float64, device=self.device)
                l.bwd_global = torch.zeros(l.cout, 2, dtype=torch.float64, device=self.device)
        self.sync_bn = True
        # the statistics now cross the ranks between the conv and the normalisation: the in-launch batch norm (whose
        # exchange stays inside one launch) is off -- rebuild the descriptors that carry it
        if not getattr(self, "plan_only", False) and any(l.fused_fwd for l in self.layers):
            self._apply_tiles()

    def _detect(self, det_thresh: float) -> None:
        L.detect(self.by_idx[75].act, self.by_idx[67].act, self.by_idx[59].act, self.B, self.S, self.num_class,
                 self.anchors.reshape(-1), self.clip_window, float(det_thresh), cfg.IOU_THRESHOLD, cfg.MAX_DETECTION,
                 self.detections, self.det_count, self.ws_det)

    def _use_half(self, h: int) -> None:
        """backbone_pair: the trainable part reads half ``h`` of the backbone outputs and the inputs of that batch"""
        self._half = int(h)
        for n, v in self._in_sets[self._half].items():
            setattr(self, n, v)
        self._build_descs()

    def sync_lanes(self) -> None:
        """overlap_tail: order the caller's stream after the side lane's open tail (the last replay's optimizer).  Cheap (one
        stream-to-stream wait, no host synchronisation); a no-op otherwise.  Everything that reads or writes training
        state outside a replay calls it first."""
        if self._tail_open and self._prog is not None:
            side = self._prog.side_stream(self.device)
            torch.cuda.current_stream().wait_stream(side)
        self._tail_open = False

    def _inputs_free_of_tail(self) -> bool:
        """overlap_tail: may the caller's stream overwrite the input tensors while the last replay's tail is still open?
        Every reader of images / labels / boxes / masks / clip windows / RoI permutations sits on the main lane or on the
        side lane IN FRONT of the mask-loss mark the main lane waits for (compute_losses) -- except the first layer's
        weight gradient, which reads the images at the very end of the side lane when layer 1 trains (stage 2).  With
        layer 1 locked the inputs are free as soon as the main lane's part of the replay is done, i.e. for anything
        the caller's stream does next."""
        return self._tail_open and self._overlap and self._prog is not None and self.by_idx[1].lock and not self.pair

    def _set_inputs(self, images, clip_window, half: int = 0, pipe_set: Optional[int] = None) -> None:
        if not self._inputs_free_of_tail():
            self.sync_lanes()
        images = torch.as_tensor(images)
        if tuple(images.shape) != (self.B, self.S, self.S, 3):
            raise ValueError("images must be [%d,%d,%d,3] NHWC (batch size and image size are baked into the plan, "
                             "as in the reference: yolo/yolo3_net_pos.py:17)" % (self.B, self.S, self.S))
        cw = self.clip_window if not self.pair else self._in_sets[half]["clip_window"]
        img = self.images
        if pipe_set is not None:
            img, cw = self._pipe_in[pipe_set]["images"], self._pipe_in[pipe_set]["clip_window"]
        img[half * self.B:(half + 1) * self.B].copy_(images.to(self.device, F32, non_blocking=True))
        cw.copy_(torch.as_tensor(clip_window).to(self.device, F32).reshape(self.B, 4))

    def forward(self, images, clip_window, det_thresh=cfg.OBJ_THRESHOLD, is_training: bool = False):
        """``sess.run(net.logits)``: returns (predictions, detections, mask_pos) with
        predictions = [yolov3_3, yolov3_2, yolov3_1] raw logits [B,g,g,3,5+C] (f32),
        detections [B,30,6], mask_pos [B,S/2,S/2,k*k] (yolo/yolo3_net_pos.py:353-357,463)."""
        # backbone_pair: the caller's images are half 0 of the 2B-image backbone pass; the trainable layers, the clip
        # windows and the outputs must be that half's too, whatever half the training loop stopped at
        prev_half = self._half
        if self.pair and prev_half != 0:
            self._use_half(0)
        try:
            self._set_inputs(images, clip_window)
            self._forward_layers(is_training)
            self._detect(float(np.asarray(det_thresh).reshape(-1)[0]))
        finally:
            if self.pair and prev_half != 0:
                self._use_half(prev_half)
        preds = [self.by_idx[i].act.view(self.B, self.by_idx[i].Ho, self.by_idx[i].Wo, 3, 5 + self.num_class)
                 for i in (75, 67, 59)]
        return preds, self.detections, self.by_idx[82].act

    def evaluation(self, images, clip_window, det_thresh=cfg.OBJ_THRESHOLD, masks_on_device: bool = False):
        """``sess.run(net.evaluation)`` (val_test, yolo/yolo3_net_pos.py:862-938): returns
        [det_box, det_mask]: per image an [n,6] array and an [n,S/2,S/2] array (scalar 0.0
        when the image has no valid detection, :933).  ``masks_on_device``: det_mask entries stay
        CUDA tensors (10 MB per image at 30 detections -- what postprocess.paste_detections takes)."""
        thr = float(np.asarray(det_thresh).reshape(-1)[0])
        if (not self.training and getattr(self, "_infer_prog", None) is not None and getattr(self, "_infer_thresh", None) == thr
                and not self.pair):
            # an inference net whose forward + filter + assembly were recorded for this threshold (build_infer_program):
            # one replay instead of ~90 launches from Python -- the same kernels on the same buffers
            self.infer(images, clip_window)
        else:
            self.forward(images, clip_window, det_thresh, is_training=False)
            Sm = self.S // 2
            if self.masks is None:
                self.masks = torch.zeros(self.B, cfg.MAX_DETECTION, Sm, Sm, dtype=F32, device=self.device)
            L.psroi_assemble(self.by_idx[82].act, self.detections, self.B, cfg.MAX_DETECTION, Sm, self.k, self.masks,
                             self.keep)
        keep = self.keep.cpu().numpy().astype(bool)
        det = self.detections.cpu().numpy()
        det_box, det_mask = [], []
        for b in range(self.B):
            det_box.append(det[b][keep[b]])
            if keep[b].any():
                m = self.masks[b][torch.from_numpy(keep[b]).to(self.device)]
                det_mask.append(m if masks_on_device else m.cpu().numpy())
            else:
                det_mask.append(np.float32(0.0))
        return [det_box, det_mask]

    def build_infer_program(self, det_thresh: float = cfg.OBJ_THRESHOLD, graph: bool = True) -> None:
        """Record inference (network + detection filter + mask assembly) into a command list
        and optionally a hipGraph; afterwards ``infer()`` is one replay.  Outputs stay on the
        device, fixed-shape: detections [B,30,6], det_count [B], masks [B,30,S/2,S/2],
        keep [B,30] (BASELINE.json config 4)."""
        Sm = self.S // 2
        if self.masks is None:
            self.masks = torch.zeros(self.B, cfg.MAX_DETECTION, Sm, Sm, dtype=F32, device=self.device)
        prog = L.CmdList()
        # a hipGraph is captured from ONE lane: the graph executor runs a captured side lane on a stream of its own choosing,
        # and whether that stream's hardware queue runs beside the launch stream's or stalls it is out of the caller's hands
        # (round 5, same box: 5.25 ms per B = 32 batch with one mapping, 7.1 ms with another; one lane: 5.42 ms always.  The
        # list replayed directly keeps its measured side lane: 5.38 ms)
        side_lane = self.use_side_lane
        if graph:
            self.use_side_lane = False
        try:
            with prog:
                self._forward_layers(False)
                self._detect(det_thresh)
                L.psroi_assemble(self.by_idx[82].act, self.detections, self.B, cfg.MAX_DETECTION, Sm, self.k, self.masks,
                                 self.keep)
        finally:
            self.use_side_lane = side_lane
        self.ws.frozen = True
        self._infer_prog = prog
        self._infer_thresh = float(det_thresh)
        self._infer_graph = None
        if graph:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                prog.run()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                prog.run()
            self._infer_graph = g

    def infer(self, images=None, clip_window=None):
        """replay the recorded inference; returns (detections, det_count, masks, keep) on device"""
        if images is not None:
            self._set_inputs(images, clip_window)
        if self._infer_graph is not None:
            self._infer_graph.replay()
        else:
            self._infer_prog.run()
        return self.detections, self.det_count, self.masks, self.keep

    # ------------------------------------------------------------------ training
    def set_batch(self, batch: Dict, half: int = 0) -> None:
        """feed_dict of Solver.train (train_yolo3_mask.py:146-149).  backbone_pair: ``half`` 0 = the batch of the next
        even step, 1 = of the odd step after it; both are set before the even step runs."""
        if half and not self.pair:
            raise L.DisyoloError("set_batch(half=1) on a net built without backbone_pair")
        if self._progs is not None and self._pipe_in is not None and not self.pair:
            # the pipelined step: the inputs of the NEXT replay (list self._parity reads set self._parity)
            q = self._parity
            if self.feed_stream is not None and torch.cuda.current_stream() == self.feed_stream:
                # feed_context(): beside the replay that is running -- the set's last reader was the replay before that one
                if self._free_valid[q]:
                    self.feed_stream.wait_event(self._ev_free[q])
                elif self._run_stream is not None:
                    self.feed_stream.wait_stream(self._run_stream)
                self._set_batch_into(batch, self._pipe_in[q], q)
                self._ev_fed[q].record(self.feed_stream)
                self._fed_pending[q] = True
            else:
                # the caller's stream, in front of the step (ordered after everything that read either set): both sets, so that
                # a batch set once and replayed (no feed per step) is what both lists read
                for qq in (q, 1 - q):
                    if self._fed_pending[qq]:          # (a feed-stream write of this set is still in flight)
                        torch.cuda.current_stream().wait_event(self._ev_fed[qq])
                        self._fed_pending[qq] = False
                    self._set_batch_into(batch, self._pipe_in[qq], qq)
                    if self._ev_free is not None:
                        # (a later feed-stream write of this set must land after this one)
                        self._ev_free[qq].record(torch.cuda.current_stream())
                        self._free_valid[qq] = True
            return
        self._set_inputs(batch["images"], batch["clip_window"], half)
        tgt = self._in_sets[half] if self.pair else {n: getattr(self, n) for n in ("labels", "true_boxes", "true_masks", "perm_det", "perm_gt")}
        self._set_labels_into(batch, tgt)

    def _set_batch_into(self, batch: Dict, tgt: Dict, pipe_set: int) -> None:
        self._set_inputs(batch["images"], batch["clip_window"], 0, pipe_set)
        self._set_labels_into(batch, tgt)

    def _set_labels_into(self, batch: Dict, tgt: Dict) -> None:
        dev = self.device
        for t, key in zip(tgt["labels"], ("yolo3", "yolo2", "yolo1")):
            t.copy_(torch.as_tensor(batch[key]).to(dev, F32).reshape(t.shape))
        tgt["true_boxes"].copy_(torch.as_tensor(batch["true_boxes"]).to(dev, F32).reshape(tgt["true_boxes"].shape))
        tm = torch.as_tensor(batch["true_masks"])
        if tm.dtype == torch.bool:
            tm = tm.view(torch.uint8)         # (0 / 1 bytes either way: no conversion pass over the 53 MB of mask planes)
        tgt["true_masks"].copy_(tm.to(dev).to(torch.uint8).reshape(tgt["true_masks"].shape))
        if batch.get("perm_det") is not None:
            tgt["perm_det"].copy_(torch.as_tensor(batch["perm_det"]).to(dev, torch.int32).reshape(tgt["perm_det"].shape))
            tgt["perm_gt"].copy_(torch.as_tensor(batch["perm_gt"]).to(dev, torch.int32).reshape(tgt["perm_gt"].shape))

    def shuffle_rois(self, generator: Optional[torch.Generator] = None) -> None:
        """tf.random_shuffle of proposals / GT boxes (yolo/yolo3_net_pos.py:781-782), host-driven
        (torch RNG).  A recorded step built with ``auto_shuffle`` does this on the device."""
        self.sync_lanes()
        B = self.B
        self.perm_det.copy_(torch.rand(B, cfg.MAX_DETECTION, device=self.device, generator=generator).argsort(dim=1))
        self.perm_gt.copy_(torch.rand(B, cfg.MAX_BOX_PER_IMAGE, device=self.device, generator=generator).argsort(dim=1))

    shuffle_seed = None   # set to an int: compute_losses reshuffles the RoI order on the device each step

    def compute_losses(self, det_thresh: float = cfg.OBJ_THRESHOLD, first_layer: int = 1) -> None:
        """forward (training mode) + detections + both losses and their gradients wrt the
        head logits / score maps (yolo/yolo3_net_pos.py:59-60)."""
        self._reg_fresh = False
        if not self._overlap_rec:
            self.sync_lanes()
        self._forward_layers(True, first_layer)
        heads = [self.by_idx[75], self.by_idx[67], self.by_idx[59]]
        side = self.use_side_lane
        if side:
            # detection filter -> RoI selection -> mask loss only feed the mask subnet's backward: side
            # lane, while the main lane finishes the mask subnet's forward, the YOLO loss and the heads'
            # backward.  The head logits (59 / 67 / 75) were produced on the side lane itself, so the
            # filter starts as soon as they exist; only the mask loss waits for the main lane (layer 82)
            L.set_lane(1)
        if self.shuffle_seed is not None:
            L.shuffle_perm(self.perm_det, self.perm_gt, self.B, int(self.shuffle_seed) & 0xffffffff, self.step_dev)
        self._detect(det_thresh)
        Sm = self.S // 2
        L.mask_rois(self.detections, cfg.MAX_DETECTION, self.true_boxes, cfg.MAX_BOX_PER_IMAGE, self.perm_det,
                    self.perm_gt, self.B, Sm, cfg.MASK_ROI_DET, cfg.MASK_ROI_GT, cfg.MASK_ROI_IOU, self.rois,
                    self.roi_count)
        if side:
            L.lane_sync(0, 1)
        m = self.by_idx[82]
        L.psroi_loss(m.act, self.true_masks, cfg.MAX_BOX_PER_IMAGE, self.rois, self.roi_count, self.B, Sm, self.k,
                     self.mask_scale, m.dx, self.mask_loss, self.ws_aux if side else self.ws)
        # the mask subnet's backward waits for THIS point of the side lane (not for the weight gradients
        # that lane picks up afterwards: a whole-lane sync there kept the main lane idle for 250-500 us)
        self._mask_mark = L.lane_mark(1) if side else -1
        if side:
            L.set_lane(0)
        L.yolo_loss([h.act for h in heads], self.labels, self.true_boxes, cfg.MAX_BOX_PER_IMAGE, self.B, self.S,
                    self.num_class, self.anchors.reshape(-1), cfg.IGNORE_THRESH,
                    (self.object_scale, self.noobject_scale, self.class_scale, self.coord_scale),
                    [h.dx for h in heads], self.losses, self.ws)
        self._mask_loss_pending = side

    def _accumulate_into(self, tgt: Layer, desc_kw: dict, dx: torch.Tensor, cin_eff: int, k: int, in_div: int,
                         final: bool = False) -> None:
        """one data-gradient conv writing (first contribution) or accumulating into tgt.grad.  ``final``: this is
        the last contribution to tgt.grad and tgt is batch-normalised -- if the conv runs the 3x3 patch kernel, its
        epilogue also emits tgt's batch-norm backward sums (bn_act_bwd then skips its column reduction)"""
        out = desc_kw.get("tmp", tgt.grad)
        first = not tgt.grad_set
        if "quad" in desc_kw and L.TUNER is None:
            ql = desc_kw["quad"]
            L.pack_quad(ql.w, ql.wq)       # (from the f32 master: the weights of THIS step, the update comes after the backward pass)
            L.dgrad_s2_quad(dx, ql.wq, tgt.grad, accumulate=not first)
            return
        res = None if (first or "tmp" in desc_kw) else tgt.grad
        d = L.make_conv_desc(dx, desc_kw["w"], out, k, 1, in_div=in_div, pads=desc_kw["pads"], out_hw=desc_kw["out_hw"],
                             residual=res)
        whole, tile_f = False, None
        if (final and L.TUNER is None and self.bn_inkernel and self.bn_inkernel_bwd and not self.sync_bn
                and self.bn_fuse_check is None and not self._shortcut_feeds_grad(tgt) and in_div == 1):
            tile_f = self._fused_tile(lambda t: L.make_conv_desc(dx, desc_kw["w"], out, k, 1, in_div=in_div, pads=desc_kw["pads"],
                                                                 out_hw=desc_kw["out_hw"], residual=res, tile=t), k, bwd=True)
            whole = tile_f is not None
        if final and L.TUNER is None and not whole and L.conv2d_bn_bwd_stats_ok(d):     # (the tuner swaps tiles under the descriptor)
            rows = L.conv2d_stats_rows(d)
            need = rows * tgt.cout * 2
            if tgt.bwd_part is None or tgt.bwd_part.numel() < need:
                tgt.bwd_part = torch.empty(need, dtype=torch.float32, device=self.device)
            tgt.bwd_part_rows = rows
            d = L.make_conv_desc(dx, desc_kw["w"], out, k, 1, in_div=in_div, pads=desc_kw["pads"], out_hw=desc_kw["out_hw"],
                                 residual=res,
                                 bn_bwd=(tgt.raw, tgt.scale, tgt.shift, tgt.mean, tgt.rstd, tgt.bwd_part, cfg.ALPHA))
        elif whole:
            # the target's WHOLE batch-norm backward inside this conv (DISYOLO_CONV_BN_BWD_FUSED): the sums are exchanged
            # within the launch and the conv writes tgt.dx; backward() then skips tgt's three batch-norm launches
            d = L.make_conv_desc(dx, desc_kw["w"], out, k, 1, in_div=in_div, pads=desc_kw["pads"], out_hw=desc_kw["out_hw"],
                                 residual=res, tile=tile_f)
            rows = L.conv2d_stats_rows(d)
            need = rows * tgt.cout * 2
            if tgt.bwd_part is None or tgt.bwd_part.numel() < need:
                tgt.bwd_part = torch.empty(need, dtype=torch.float32, device=self.device)
            if tgt.csync_bwd is None:
                tgt.csync_bwd = L.cluster_sync_buffer(tgt.cout, self.device)
            d = L.make_conv_desc(dx, desc_kw["w"], tgt.dx, k, 1, in_div=in_div, pads=desc_kw["pads"], out_hw=desc_kw["out_hw"],
                                 residual=res, tile=tile_f,
                                 bn_bwd=(tgt.raw, tgt.scale, tgt.shift, tgt.mean, tgt.rstd, tgt.bwd_part, cfg.ALPHA),
                                 bn_bwd_fused=dict(dgamma=tgt.dgamma, dbeta=tgt.dbeta, sync=tgt.csync_bwd))
            tgt.fused_bwd = True
        L.conv2d_fwd(d)

    def check_cluster_sync(self) -> None:
        """raise if a bounded wait of an in-launch exchange ever gave up (its launch then ran on with wrong sums): call at a
        synchronisation point -- bench.py does after its timed regions, Solver at every loss fetch"""
        for l in self.layers:
            for name, buf in (("forward", l.csync), ("backward", l.csync_bwd)):
                if buf is not None:
                    code = L.cluster_sync_error(buf, l.cout)
                    if code:
                        raise L.DisyoloError("conv%d: the in-launch batch-norm exchange (%s) timed out (code %#x): another launch of "
                                             "that kind was running on this device, or the grid was not resident; the results of "
                                             "this run are wrong (DISYOLO_BN_INKERNEL=0 switches the feature off)" % (l.idx, name, code))

    def _shortcut_feeds_grad(self, tgt: Layer) -> bool:
        """tgt is a residual layer whose output gradient also goes to its shortcut's source (res_conv_bn): that copy rides
        on the separate batch-norm backward's apply pass"""
        return tgt.shortcut is not None and self.by_idx[tgt.shortcut].grad is not None

    def _final_writers(self, visit) -> Dict[int, Tuple[int, str]]:
        """layer idx -> (idx of the layer whose backward makes its output gradient final, how: "direct" = a
        data-gradient conv straight into it, "up" = through the 2x upsampling, "add" = a shortcut's add)"""
        last: Dict[int, Tuple[int, str]] = {}
        for l in visit:
            if l.lock:
                continue
            if l.kind != "lin" and l.shortcut is not None and self.by_idx[l.shortcut].grad is not None:
                last[l.shortcut] = (l.idx, "add")
            for mode, tgt, _ in l.dgrad_descs:
                last[tgt.idx] = (l.idx, mode)
        return last

    def backward_order(self) -> List[Layer]:
        """the order backward() visits the layers in: the three detection heads (75,74 / 67,66 / 59,58)
        first, then the rest in descending order (any topological order of the reversed graph is valid)"""
        heads = [self.by_idx[i] for i in (75, 74, 67, 66, 59, 58)]
        return heads + [l for l in reversed(self.layers) if l.idx not in self.HEAD_LAYERS]

    def backward(self, on_layer_done=None, sweep: bool = False) -> None:
        """TF autodiff of total_loss restated layer by layer in reverse order.  Gradients of
        the trainable variables land in ``grad_arena``.  ``on_layer_done(layer)`` is called
        after a layer's parameter gradients are enqueued (used to overlap the RCCL
        all-reduce with the rest of the backward pass).  ``sweep`` (train_step, recorded steps): the
        optimizer update of an arena slice is issued as soon as the slice's gradients are final
        (optimizer_step() then only sweeps what is left and finishes the step)."""
        B = self.B
        for l in self.layers:
            l.grad_set = False
        # any topological order of the reversed graph is valid.  The three detection heads
        # (75,74 / 67,66 / 59,58) depend on the YOLO loss only, so they go first while the side
        # lane still runs the detection filter and the mask loss; the mask subnet follows.
        visit = self.backward_order()
        order = [l for l in visit if not l.lock]
        fuse_bn = os.environ.get("DISYOLO_BN_FUSE", "1") != "0" and not self.sync_bn
        final_of = self._final_writers(visit) if fuse_bn else {}
        for l in self.layers:
            l.bwd_part_rows = 0
            l.fused_bwd = False
        # (data parallelism with a cut list: the sweep must follow the bucket's all-reduce -- it stays in optimizer_step;
        # with the exchange in the list the slice's all-reduce and its sweep go to the exchange lane together)
        inl = self.dp is not None and self.dp.inlist
        overlap_opt = sweep and (self.dp is None or inl) and self.n_params > 0 and os.environ.get("DISYOLO_OPT_OVERLAP", "1") != "0"
        if overlap_opt:
            if self.opt_chunks is None:
                self._plan_opt_chunks()
            self._opt_swept = set()
            self._opt_done = [set() for _ in self.opt_chunks]
            self._dp_pending = []        # (slice, mark of its collective on the exchange lane, layer position): exchanged, not yet swept
            self._dp_sweep_delay = int(os.environ.get("DISYOLO_DP_SWEEP_DELAY", "2"))
            # where a slice's sweep runs: "comm" = on the exchange lane right behind the slice's collective (default: a hop
            # between lanes costs ~15 us, profiles/r05_hw_queues.txt, and the side lane is nearly as busy as the main one),
            # "side" = back on the side lane DISYOLO_DP_SWEEP_DELAY layers later
            self._dp_sweep_on_comm = self._dp_sweep_lane() != 1
        group = max(1, int(os.environ.get("DISYOLO_WGRAD_GROUP", "3")))
        pending: list = []            # (layer, dx, ld, M, side, pos) whose weight gradients wait for the group's edge

        def flush():
            if not pending:
                return
            side_g = pending[0][4]
            mark = L.lane_mark(0) if side_g else None         # ONE record on the main lane for the whole group
            waited = False
            for (pl, pdx, pld, pM, pside, ppos) in pending:
                if pside:
                    if not waited:
                        L.lane_wait(mark, 1)
                        waited = True
                    L.set_lane(1)
                # scratch of the lane the gradient runs on: the side lane's weight gradients own ws_aux; one kept on the main lane
                # (tail_on_main) must not share it -- the side lane may still be inside an earlier layer's slabs (found in round 6:
                # with DISYOLO_TAIL_MAIN > 0 the loss differed from run to run)
                wsl = self.ws_aux if pside else self.ws
                if pl.kind == "lin":
                    L.colsum(pl.dx, pl.dbias, pM, L.GRAD_LD, pl.cout, wsl)   # bias gradient
                if pl.idx == 1:
                    if pl.cout == 32 and os.environ.get("DISYOLO_FIRST_WGRAD_MFMA", "1") != "0":
                        # the first layer's own kernel: taps as the M axis of the MFMA, the f32 image rounded to bf16 on
                        # its way into LDS (csrc/conv_wgrad.hip, conv_first_wgrad_mfma_kernel)
                        L.conv_first_wgrad(self.images, pl.dx, pl.dw, wsl)
                    else:
                        # through the im2col kernel: bf16 image padded to 8 channels, K = 9*8 rows of which 27 are real
                        L.image_pad8(self.images, self._img8)
                        L.conv2d_wgrad(self._wgrad1_desc, pl.dx, pl.cout, self._dw8, wsl)
                        L.copy2d_f32(self._dw8, pl.dw, 9, 3 * pl.cout, 8 * pl.cout, 3 * pl.cout)
                elif os.environ.get("DISYOLO_EXP_SKIP_WGRAD") not in ("1", "2"):     # (experiment: the step without its weight gradients)
                    L.conv2d_wgrad(pl.wgrad_desc, pdx, pld, pl.dw, wsl)
                if pside:
                    L.set_lane(0)
                if self._overlap_rec:
                    for src, reader in self._xstep_reader.items():
                        if reader == pl.idx and self._xstep_slot[src] in self._xstep_waits.values():     # (only slots somebody waits on)
                            L.lane_mark_slot(1 if pside else 0, self._xstep_slot[src])
                if overlap_opt and pside:
                    # the optimizer sweep of an arena slice (+ the re-pack of its layers) as soon as the slice's weight
                    # gradients are final, on the side lane behind them -- not at the end of the critical chain.  It rewrites
                    # the bf16 operands the data-gradient convs of these layers read: they are all behind the group's edge,
                    # which the side lane has waited for
                    ci = self._opt_chunk_of[pl.idx]
                    self._opt_done[ci].add(pl.idx)
                    if self._opt_done[ci] == self.opt_chunks[ci]["members"]:
                        if inl:
                            # data parallel: the slice's gradients are summed over the ranks first -- the collective goes to
                            # the exchange lane (behind the side lane's weight gradients of the slice; the side lane itself
                            # goes on with the next layers' weight gradients), and the sweep of a slice is issued a few
                            # layers later (DISYOLO_DP_SWEEP_DELAY), on the side lane like the single-GPU step's: its
                            # collective has had those layers' time on the links before the side lane sits behind it
                            L.lane_wait(L.lane_mark(1), L.COMM_LANE)
                            L.set_lane(L.COMM_LANE)
                            ch = self.opt_chunks[ci]
                            self.dp.exchange_inlist(ci, ch["off"], ch["cnt"])
                            if self._dp_sweep_on_comm:
                                # ... and the sweep right behind it on the exchange lane (lowest stream priority, like the side
                                # lane the single-GPU step sweeps on): the side lane never waits for the links; its mark sits
                                # behind the group's main-lane edge, so the re-pack is safe
                                self._sweep_chunk(ci, 1.0 / self.dp.world_size)
                            else:
                                self._dp_pending.append((ci, L.lane_mark(L.COMM_LANE), ppos))
                            L.set_lane(0)
                        else:
                            L.set_lane(1)
                            self._sweep_chunk(ci, 1.0)
                            L.set_lane(0)
                if overlap_opt and inl:
                    # the sweep of a slice follows its collective DISYOLO_DP_SWEEP_DELAY layers later, on the side lane
                    while self._dp_pending and ppos - self._dp_pending[0][2] >= self._dp_sweep_delay:
                        self._dp_sweep_oldest(1, main_waited=pside)
                if on_layer_done is not None:
                    on_layer_done(pl)
            pending.clear()

        for l in visit:
            pos = order.index(l) if not l.lock else -1
            if l.idx == 82 and getattr(self, "_mask_loss_pending", False):
                L.lane_wait(self._mask_mark, 0)          # dscore comes from the side lane
                self._mask_loss_pending = False
            if l.lock:
                # locked layers still pass gradients through their residual add only in
                # stage 2; in stage 1 nothing upstream is trainable
                continue
            M = B * l.Ho * l.Wo
            if l.kind == "lin":
                dx, ld = l.dx, L.GRAD_LD
            else:
                if not l.grad_set:
                    raise L.DisyoloError("layer %d received no gradient" % l.idx)
                # a residual layer hands its output gradient on to the shortcut's source (res_conv_bn, :148-151): that
                # copy / add rides on the batch-norm backward's apply pass, which reads l.grad anyway
                sc = self.by_idx[l.shortcut] if l.shortcut is not None else None
                if sc is not None and sc.grad is None:
                    sc = None
                fuse_sc = sc is not None and os.environ.get("DISYOLO_SHORTCUT_FUSE", "1") != "0"
                kw = dict(shortcut_grad=sc.grad, shortcut_accumulate=sc.grad_set) if fuse_sc else {}
                if l.fused_bwd:
                    pass        # l.dx, l.dgamma, l.dbeta came out of the data-gradient conv that made l.grad final (_accumulate_into)
                elif self.sync_bn:
                    # (sum g, sum g*xhat) of this rank -> the same over all ranks -> dx; dgamma / dbeta stay local
                    L.bn_bwd_reduce(l.grad, l.raw, l.scale, l.shift, l.mean, l.rstd, M, l.cout, l.bwd_local, self.ws, cfg.ALPHA)
                    L.copy2d_f32(l.bwd_local.view(torch.float32), l.bwd_global.view(torch.float32), 1, 4 * l.cout,
                                 4 * l.cout, 4 * l.cout)
                    self._sync_sums(l.bwd_global)
                    L.bn_bwd_apply_sums(l.grad, l.raw, l.scale, l.shift, l.mean, l.rstd, l.bwd_local, l.bwd_global,
                                        M * self.dp.world_size, l.dx, l.dgamma, l.dbeta, M, l.cout, self.ws, cfg.ALPHA, **kw)
                elif l.bwd_part_rows:
                    # the patch conv that made l.grad final left the batch-norm backward sums, one row per patch
                    ref = None
                    if self.bn_fuse_check is not None and self._rec is None:
                        # debug mode: the plain three-kernel backward on the same gradient first, into scratch
                        ref = (torch.empty_like(l.dx), torch.empty_like(l.dgamma), torch.empty_like(l.dbeta))
                        L.bn_act_bwd(l.grad, l.raw, l.scale, l.shift, l.mean, l.rstd, ref[0], ref[1], ref[2], M, l.cout,
                                     self.ws, cfg.ALPHA)
                    L.bn_act_bwd_partials(l.grad, l.raw, l.scale, l.shift, l.mean, l.rstd, l.dx, l.dgamma, l.dbeta, M,
                                          l.cout, l.bwd_part, l.bwd_part_rows, self.ws, cfg.ALPHA, **kw)
                    if ref is not None:
                        torch.cuda.synchronize()

                        def rel(a, b):
                            a, b = a.double().flatten(), b.double().flatten()
                            return float((a - b).norm() / (b.norm() + 1e-30))
                        self.bn_fuse_check.append({"layer": l.idx, "rows": l.bwd_part_rows, "shape": (l.Ho, l.Wo, l.cout),
                                                   "dx": rel(l.dx, ref[0]), "dgamma": rel(l.dgamma, ref[1]),
                                                   "dbeta": rel(l.dbeta, ref[2]),
                                                   "finite": bool(torch.isfinite(l.dx.float()).all())})
                else:
                    L.bn_act_bwd(l.grad, l.raw, l.scale, l.shift, l.mean, l.rstd, l.dx, l.dgamma, l.dbeta, M, l.cout,
                                 self.ws, cfg.ALPHA, **kw)
                dx, ld = l.dx, l.cout
                if sc is not None:
                    if not fuse_sc:
                        L.add_bf16(l.grad, sc.grad, accumulate=sc.grad_set)
                    sc.grad_set = True
            # the weight gradient is off the critical chain (dx -> dgrad -> next layer's BN
            # backward): in a recorded step it runs on the side lane, overlapping the small
            # latency-bound BN kernels of the following layers
            # ... except for the last few layers of the pass: nothing is left on the main lane to
            # overlap with, the step would only wait for the side lane's backlog to drain
            # (data parallelism with a CUT list: every weight gradient stays on the side lane -- the bucket all-reduce is ordered
            # after that lane only.  With the exchange in the list the slices whose last weight gradient ran on the main lane are
            # exchanged and swept in optimizer_step, whose exchange lane waits for the main lane, which has joined the side lane)
            tail = self.tail_on_main if (self.dp is None or (inl and os.environ.get("DISYOLO_DP_TAIL_MAIN", "1") != "0")) else 0
            side = self.use_side_lane and (pos < len(order) - tail) and os.environ.get("DISYOLO_EXP_SKIP_WGRAD") != "2"
            # enqueue order = host order: the data-gradient convs (critical chain, main lane) are issued before the
            # weight gradient, which only needs dx.  The edge to the side lane is an event RECORDED ON THE MAIN LANE, and a
            # record costs the recording stream ~3 us (tools/micro/event_cost.hip: a chain of 10-us kernels, 11.0 -> 13.8 us
            # per kernel with a record behind each, 12.0 with one per three): one edge per layer was ~0.1 ms of a 4.1-ms
            # step.  So the weight gradients of DISYOLO_WGRAD_GROUP consecutive layers share ONE edge, recorded behind the
            # group's data-gradient convs; the side lane lags a layer or two more, which nothing waits for.
            for mode, tgt, kw in l.dgrad_descs:
                if mode == "direct":
                    final = (final_of.get(tgt.idx) == (l.idx, "direct") and not tgt.lock and tgt.kind != "lin"
                             and tgt.cout % 8 == 0 and tgt.raw is not None)
                    self._accumulate_into(tgt, kw, dx, l.cin, l.k, l.stride, final)
                    tgt.grad_set = True
                else:
                    self._accumulate_into(tgt, kw, dx, l.cin, l.k, l.stride)
                    up = tgt
                    L.upsample2x_bwd(kw["tmp"], up.grad, B, l.H, l.W, up.cout, 0, up.cout, accumulate=up.grad_set)
                    up.grad_set = True
            if pending and pending[-1][4] != side:
                flush()
            pending.append((l, dx, ld, M, side, pos))
            if not side or len(pending) >= group:
                flush()
        flush()
        if not self._overlap_rec:
            L.lane_sync(1, 0)

    def _apply_tiles(self) -> None:
        """rebuild everything that depends on a tile choice: batch-norm partial-sum buffers
        (one row per M tile) and the conv descriptors"""
        for l in self.layers:
            if l.stats is not None and l.idx > 1:
                l.stats = None
        self._build_descs()
        if self.training:
            self._build_dgrad_descs()

    def autotune(self, reps: int = 3, candidates=None, det_thresh: float = cfg.OBJ_THRESHOLD, cache=None) -> dict:
        """Pick the conv tile per layer shape by timing the candidates inside the real layer
        sequence (forward + backward, eager launches) on the batch currently set; training
        state is saved and restored around it.  Call before ``build_program``.  Returns
        {shape key: tile code} (0 = launcher heuristic kept).  ``cache``: a JSON file the picks
        are loaded from when present (no timing passes) and written to otherwise."""
        if self._prog is not None:
            raise L.DisyoloError("autotune() must run before build_program()")
        if cache and os.path.exists(cache):
            with open(cache) as f:
                picks = {tuple(json.loads(k)): int(v) for k, v in json.load(f).items()}
            L.TUNED.clear()
            L.TUNED.update({k: v for k, v in picks.items() if v})
            self._apply_tiles()
            self.tuned = picks
            return picks
        state = {k: v.clone() for k, v in self.state_dict().items()} if self.training else None
        # batch-norm partial sums are written per M tile: during tuning size them for the
        # smallest BM of any candidate (results are not used, only in-bounds)
        saved_stats = {}
        for l in self.layers:
            if l.stats is not None and l.idx > 1:
                saved_stats[l.idx] = (l.stats, l.stats_rows)
                l.stats = torch.zeros(-(-self.B * l.Ho * l.Wo // 64), l.cout, 2, dtype=F32, device=self.device)
        if saved_stats:
            self._build_descs()
        tuner = L.ConvTuner(candidates or L.TUNE_CANDIDATES)
        L.TUNED.clear()
        L.TUNER = tuner
        try:
            for cand in (0,) + tuner.candidates:
                tuner.current = cand
                for _ in range(reps):
                    if self.training:
                        self.compute_losses(det_thresh)
                        self.backward()
                    else:
                        self._forward_layers(False)
            picks = tuner.commit()
        finally:
            L.TUNER = None
            torch.cuda.synchronize()
        self._apply_tiles()
        if self.training:
            self.load_state_dict(state)
        self.tuned = picks
        if cache:
            with open(cache, "w") as f:
                json.dump({json.dumps(list(k)): v for k, v in picks.items()}, f)
        return picks

    @property
    def step_count(self) -> int:
        self.sync_lanes()
        return int(self.step_dev.item())

    # ---- optimizer: Adam as sweeps over slices of the arena ------------------------------
    def _plan_opt_chunks(self) -> None:
        """Slices of the regularised region [0, n_decay) in layer order, a few million variables each
        (DISYOLO_OPT_CHUNK_M, default 8).  A slice can be swept as soon as the weight gradients of its
        layers are final; each slice also owns the re-pack of its layers' bf16 operands and a fixed
        range of the l2 partial sums (so the value of the l2 term does not depend on when it ran)."""
        from .dp import plan_buckets
        spans = []
        for l in self.layers:
            if l.lock:
                continue
            o, c = self.arena_slices[var_name(l.idx, "weights")]
            if l.kind == "lin":
                ob, cb = self.arena_slices[var_name(l.idx, "biases")]
                assert ob == o + c
                c += cb
            spans.append((l.idx, o, c))
        for (_, o, c), (_, o2, _) in zip(spans, spans[1:]):
            assert o + c == o2, "arena must be contiguous in layer order"
        assert not spans or (spans[0][1] == 0 and spans[-1][1] + spans[-1][2] == self.n_decay)
        elems = int(float(os.environ.get("DISYOLO_OPT_CHUNK_M", "8")) * 1e6)
        self.opt_chunks = []
        self._opt_chunk_of = {}
        parts = 0
        by = {l.idx: l for l in self.layers}
        for _, off, cnt in plan_buckets(spans, elems):
            members = {idx for idx, o, c in spans if off <= o and o + c <= off + cnt}
            jobs = [(by[i].w, by[i].wp, by[i].wdg, by[i].k, by[i].cin, by[i].cout, by[i].cout_pad)
                    for i in sorted(members) if i > 1]
            ch = {"off": off, "cnt": cnt, "members": members, "parts_off": parts,
                  "nparts": L.adam_sweep_parts(cnt), "pack": L.PackTable(jobs, self.device) if jobs and not self.plan_only else None}
            parts += ch["nparts"]
            for i in members:
                self._opt_chunk_of[i] = len(self.opt_chunks)
            self.opt_chunks.append(ch)
        self._opt_nparts = parts
        self._opt_parts = None if self.plan_only else torch.zeros(max(parts, 1), dtype=torch.float32, device=self.device)
        self._opt_swept = set()
        self._opt_done = [set() for _ in self.opt_chunks]

    def _sweep_chunk(self, ci: int, grad_scale: float) -> None:
        """Adam over slice ci + the re-pack of its layers"""
        ch = self.opt_chunks[ci]
        o, c = ch["off"], ch["cnt"]
        L.adam_sweep(self.arena[o:o + c], self.grad_arena[o:o + c], self.adam_m[o:o + c], self.adam_v[o:o + c], c, c,
                     self.lr_dev, cfg.ADAM_BETA1, cfg.ADAM_BETA2, cfg.ADAM_EPSILON, self.l2, self.step_dev, grad_scale,
                     self._opt_parts[ch["parts_off"]:ch["parts_off"] + ch["nparts"]])
        if ch["pack"] is not None:
            ch["pack"].run()
        self._opt_swept.add(ci)

    def _dp_sweep_lane(self) -> int:
        """data parallel, exchange in the list: the lane a slice's optimizer sweep runs on.  DISYOLO_DP_SWEEP_LANE = "comm"
        (default): the exchange lane itself, right behind the slice's collective -- the side lane never waits for the links;
        that lane has the lowest stream priority (DISYOLO_LANE3_LOW=0: normal), so the HBM-bound sweeps do not compete with
        the main lane, as on the single-GPU step's side lane.  One RCCL rank, same box, ms per step: plain 4.16-4.18 |
        comm + low priority 4.17-4.18 | comm at normal priority 4.22 | "side" (back on the side lane
        DISYOLO_DP_SWEEP_DELAY layers later) 4.20 | a third, low-priority lane for the sweeps behind a normal-priority
        exchange lane 4.15-4.18 against 4.12 for comm + low on another box (one more 15-us hop): not kept."""
        return 1 if os.environ.get("DISYOLO_DP_SWEEP_LANE", "comm") == "side" else L.COMM_LANE

    def _dp_sweep_oldest(self, lane: int, main_waited: bool = False) -> None:
        """data parallel, exchange in the list: the optimizer sweep (+ re-pack) of the slice whose collective was issued
        first, on ``lane`` behind that collective and behind the main lane up to here (the re-pack rewrites operands the
        main lane's data-gradient convs read)"""
        ci, mk = self._dp_pending.pop(0)[:2]
        L.lane_wait(mk, lane)
        if lane != 0 and not main_waited:         # (main_waited: the lane already sits behind an edge of the main lane that is late enough)
            L.lane_wait(L.lane_mark(0), lane)
        prev = L.CURRENT_LANE
        L.set_lane(lane)
        self._sweep_chunk(ci, 1.0 / self.dp.world_size)
        L.set_lane(prev)

    def optimizer_step(self, grad_scale: float = 1.0) -> None:
        """tf.train.AdamOptimizer(1e-4).minimize (train_yolo3_mask.py:55) over the arena; the
        l2 regulariser's gradient (l2*w) is folded in for weights and biases.  The step count
        lives on the device so the whole step can be replayed without host state.  Slices that
        backward() already swept (their gradients were final early) are skipped here; the finish
        produces the l2 term of total_loss and advances the step count."""
        if self.n_params:
            if self.opt_chunks is None:
                self._plan_opt_chunks()
            nt = self.n_params - self.n_decay       # batch-norm gamma / beta: not regularised
            inl = self.dp is not None and self.dp.inlist
            on_comm = inl and self._dp_sweep_lane() != 1
            cur = L.CURRENT_LANE
            if inl and on_comm:
                # what is left of the exchange and of the optimizer on the exchange lane, in order: each remaining slice's
                # collective and its sweep, gamma / beta, the finish; this lane then waits for the exchange lane once
                sl = L.COMM_LANE
                L.lane_wait(L.lane_mark(cur), L.COMM_LANE)
                L.set_lane(L.COMM_LANE)
                for ci in range(len(self.opt_chunks)):
                    if ci not in self._opt_swept:
                        ch = self.opt_chunks[ci]
                        self.dp.exchange_inlist(ci, ch["off"], ch["cnt"])
                        self._sweep_chunk(ci, grad_scale)
                if nt > 0:
                    self.dp.exchange_inlist("tail", self.n_decay, nt)
            elif inl:
                # the collectives of everything not exchanged yet (the slices that became final last, gamma / beta) on the
                # exchange lane, behind what this lane has seen; the sweeps follow on this lane, oldest collective first
                pend = {p[0] for p in self._dp_pending}
                L.lane_wait(L.lane_mark(cur), L.COMM_LANE)
                L.set_lane(L.COMM_LANE)
                for ci in range(len(self.opt_chunks)):
                    if ci not in self._opt_swept and ci not in pend:
                        ch = self.opt_chunks[ci]
                        self.dp.exchange_inlist(ci, ch["off"], ch["cnt"])
                        self._dp_pending.append((ci, L.lane_mark(L.COMM_LANE), 0))
                if nt > 0:
                    self.dp.exchange_inlist("tail", self.n_decay, nt)
                tail_mark = L.lane_mark(L.COMM_LANE)
                L.set_lane(cur)
                while self._dp_pending:
                    self._dp_sweep_oldest(cur)
                L.lane_wait(tail_mark, cur)
            for ci in range(len(self.opt_chunks)):
                if ci not in self._opt_swept:
                    self._sweep_chunk(ci, grad_scale)
            if nt > 0:
                o = self.n_decay
                L.adam_sweep(self.arena[o:], self.grad_arena[o:], self.adam_m[o:], self.adam_v[o:], nt, 0, self.lr_dev,
                             cfg.ADAM_BETA1, cfg.ADAM_BETA2, cfg.ADAM_EPSILON, self.l2, self.step_dev, grad_scale, None)
            L.adam_finish(self.step_dev, self._opt_parts if self.n_decay else None, self._opt_nparts if self.n_decay else 0,
                          self.l2, self.reg_loss if self.n_decay else None,
                          record=(self.losses, self.mask_loss, None if self.n_decay else self.reg_loss, self.loss_ring))
            if inl and on_comm:
                L.set_lane(cur)
                L.lane_sync(sl, cur)
            self._opt_swept = set()
            self._opt_done = [set() for _ in self.opt_chunks]
            self._dp_pending = []
            self._reg_fresh = True

    LOSS_RING = 1024

    def step_losses(self, first: int, count: int) -> np.ndarray:
        """total_loss of the optimizer steps first .. first+count-1 (0-based, counted like ``step_count``) as the
        finish filed them on the device -- bit for bit what total_loss() returned after each of those steps.  Lets a
        training loop accumulate the loss of every step, as the reference does with the value sess.run hands back
        (train_yolo3_mask.py:216-218), while fetching from the device once per ``count`` <= LOSS_RING steps instead of
        once per step (each fetch joins an overlapped tail)."""
        if count < 0 or count > self.LOSS_RING:
            raise L.DisyoloError("step_losses: at most %d steps are kept" % self.LOSS_RING)
        done = self.step_count         # (joins)
        if first < 0 or first + count > done or done - first > self.LOSS_RING:
            raise L.DisyoloError("step_losses: steps %d..%d are not in the ring (steps done: %d, ring %d)"
                                 % (first, first + count - 1, done, self.LOSS_RING))
        ring = self.loss_ring.cpu().numpy()
        return np.asarray([ring[(first + i) % self.LOSS_RING] for i in range(count)], np.float32)

    def repack(self) -> None:
        """bf16 MFMA operands of every trainable layer from the f32 masters"""
        self.sync_lanes()
        if self.opt_chunks is None:
            self._plan_opt_chunks()
        for ch in self.opt_chunks:
            if ch["pack"] is not None:
                ch["pack"].run()

    def total_loss(self) -> torch.Tensor:
        """conf + class + coord + mask + l2 term as a device scalar (tf.losses.get_total_loss,
        yolo/yolo3_net_pos.py:61).  Valid after compute_losses(); the l2 term is evaluated on
        the current weights (inside a recorded step: the pre-update weights, like TF)."""
        self.sync_lanes()
        if self.n_decay and not self._reg_fresh:
            # called between compute_losses() and the optimizer: evaluate the l2 term now; after
            # optimizer_step() / a recorded step reg_loss already holds it (written by the Adam sweep)
            L.l2_loss(self.arena, self.n_decay, self.l2, self.reg_loss, self.ws)
        return self.losses[7] + self.mask_loss[0] + self.reg_loss[0]

    # ---- recorded step: one C call (or one hipGraph launch) per iteration -------------
    def build_program(self, det_thresh: float = cfg.OBJ_THRESHOLD, graph: bool = False,
                      pipeline_backbone: bool = False, overlap_tail: bool = False) -> None:
        """Record forward + losses + backward + Adam + re-pack into a command list
        (csrc/runtime.hip).  With data parallelism the list is cut where a gradient bucket
        becomes final so the RCCL all-reduces are issued between segments.

        ``pipeline_backbone`` (stage 1 only): the locked backbone's forward pass does not depend
        on anything the step updates, so the recorded step computes it for the NEXT batch's
        images on a third lane while it runs heads/losses/backward/Adam for the current batch
        (whose backbone pass ran during the previous step; outputs double-buffered).  Call
        ``set_batch`` with labels of batch t and images of batch t+1, after ``prime_pipeline``.
        """
        if not self.training:
            raise L.DisyoloError("build_program on a YOLONet built with training=False")
        self.sync_lanes()            # (a previous list's tail may still run: join before that list is dropped)
        # ``overlap_tail`` (single GPU, list executor): the step does not join its side lane at the end.  The side lane still
        # owes the optimizer sweeps of the slices that became final last (three quarters of the stage-1 parameters sit in the
        # layers the backward pass reaches last) and the last weight gradients -- 0.15 ms during which the main lane was idle;
        # the next replay starts its locked-backbone forward right away and waits, per tensor, for the side-lane work that
        # still reads it (the weight gradients of the layers fed by backbone outputs) and, in front of the first trainable
        # layer, for the optimizer + re-pack.  Same kernels, same order per lane: results bit-identical.  train_step(want_loss=
        # True), total_loss(), forward(), set_batch(), state_dict() ... join first (sync_lanes()).
        if overlap_tail and (graph or pipeline_backbone or self.pair or (self.dp is not None and not self.dp.inlist)
                             or not self.use_side_lane):
            raise L.DisyoloError("overlap_tail needs the two-lane list executor (no graph / pipeline / pair; data parallelism "
                                 "only with the exchange in the list)")
        self._overlap = bool(overlap_tail)
        if self._overlap:
            self._plan_overlap()
        if self.pair:
            # two lists: the even step = the backbone on both batches + the trainable part on the first half, the odd
            # step = the trainable part on the second half; run_program alternates
            if graph or pipeline_backbone or self.dp is not None:
                raise L.DisyoloError("backbone_pair uses the single-GPU list executor")
            self._progs = []
            for h in (0, 1):
                self._use_half(h)
                self._progs.append(self._record_step(det_thresh, None))
            self._use_half(0)
            self._parity = 0
            self.ws.frozen = True
            self.ws_aux.frozen = True
            self._prog, self._prog_marks, self._bwd_end = self._progs[0]
            return
        if pipeline_backbone:
            if graph:
                raise L.DisyoloError("pipeline_backbone uses the list executor, not a hipGraph")
            if "DISYOLO_TAIL_MAIN" not in os.environ and (self.dp is None or self.dp.inlist):
                # the pipelined step joins its lanes at the end of every replay: the weight gradients of the LAST two layers of
                # the backward pass stay on the main lane, which would only wait for them (tail 0 / 1 / 2 / 3 / 4 / 6 / 8 layers:
                # 3.82 / 3.80 / 3.76 / 3.76 / 3.81 / 3.85 / 3.94 ms per step, profiles/r06_backbone_pipeline.txt)
                self.tail_on_main = 2
            self._setup_pipeline()
            self._progs = []
            for q in (0, 1):
                prog, marks, bwd_end = self._record_step(det_thresh, q)
                self._progs.append((prog, marks, bwd_end))
            self._pick_feed_stream()
            self._use_parity(0)
            self.ws.frozen = True
            self.ws_aux.frozen = True
            self._prog, self._prog_marks, self._bwd_end = self._progs[0]
            return
        self._overlap_rec = self._overlap
        try:
            prog, marks, self._bwd_end = self._record_step(det_thresh, None)
        finally:
            self._overlap_rec = False
        self.ws.frozen = True
        self.ws_aux.frozen = True
        self._prog, self._prog_marks = prog, marks
        if graph:
            if self.dp is not None:
                raise L.DisyoloError("hipGraph replay is only wired for the single-GPU step")
            # one warm-up replay outside capture (first-launch module loads are not capturable),
            # with every piece of training state saved and restored around it
            state = [self.arena, self.adam_m, self.adam_v, self.step_dev] + \
                    [t for l in self.layers if l.kind != "lin" for t in (l.mm, l.mv)]
            saved = [t.clone() for t in state]
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                prog.run()
            torch.cuda.current_stream().wait_stream(side)
            for t, v in zip(state, saved):
                t.copy_(v)
            self.refresh_weights()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                prog.run()
            self._graph = g

    def _record_step(self, det_thresh: float, parity):
        """one recorded step; parity None = plain, 0/1 = pipelined (consumes backbone outputs of
        that parity, produces the other)"""
        inl = self.dp is not None and self.dp.inlist
        if self.dp is not None and not inl:
            # the RCCL all-reduces are issued on the side lane: keep it at normal stream priority.  The lanes are streams of a
            # process-wide pool whose priority is fixed when a lane is first used: the setting only helps if lane 1 does not
            # exist yet -- say so when it is too late (a net that has recorded or tuned before enable_data_parallel)
            os.environ.setdefault("DISYOLO_LANE1_LOW", "0")
            if not getattr(self, "plan_only", False):
                for ln in L.lanes_report():
                    if ln.startswith("lane 1:") and "priority 0," not in ln and os.environ.get("DISYOLO_LANE1_LOW") == "0":
                        print("disyolo: the cut-list data-parallel step wants lane 1 at normal priority, but the lane already exists "
                              "(%s): its bucket all-reduces queue behind a lowest-priority stream -- enable data parallelism before "
                              "anything records or tunes, or use the in-list exchange (RCCL process group)" % ln, file=sys.stderr)
        prog = L.CmdList()
        marks = []
        if self.sync_bn and parity is not None:
            raise L.DisyoloError("SyncBN is not wired for the pipelined-backbone step")
        self._rec = (prog, marks) if (self.dp is not None and not inl) else None
        with prog:
            first = 1
            if parity is not None:
                self._use_parity(parity)
                first = self._pipe_P + 1
            if self.pair and self._half == 1:
                first = self._pair_P + 1         # (the even step ran the backbone for this half too)
            pipe_early = parity is not None and os.environ.get("DISYOLO_PIPE_EARLY", "0") == "1"

            def next_backbone():
                # the next batch's backbone: lowest-priority lane (lane 2)
                self._use_parity(1 - parity)
                self.images = self._pipe_in[parity]["images"]     # (fed before THIS replay: labels of batch t, images of t + 1)
                lane = int(os.environ.get("DISYOLO_PIPE_LANE", "2"))
                L.lane_sync(0, lane)
                L.set_lane(lane)
                self._forward_prefix(self._pipe_P, True)
                L.set_lane(0)
                self._use_parity(parity)
            if pipe_early:
                next_backbone()      # (experiment: from the start of the step, beside the heads' forward pass as well)
            started = [False]
            fwd_after = int(os.environ.get("DISYOLO_PIPE_FWD_AFTER", "0")) if (parity is not None and not pipe_early) else 0
            if fwd_after:
                # (experiment: from behind forward layer DISYOLO_PIPE_FWD_AFTER of the trainable part, a main-lane layer)
                def early_start():
                    started[0] = True
                    next_backbone()
                self._fwd_hook = (fwd_after, early_start)
            self.compute_losses(det_thresh, first)
            self._fwd_hook = None
            pipe_after = int(os.environ.get("DISYOLO_PIPE_AFTER", "0")) if (parity is not None and self.dp is None) else 0
            if parity is not None and not pipe_early and pipe_after == 0 and not started[0]:
                # ... started once the trunk's forward is done, so it fills what the backward pass leaves of the CUs
                next_backbone()
            if self.dp is not None and not inl:
                self.dp.begin_step()

                def mark(l):
                    # weight / bias gradients are produced on the side lane: the bucket's
                    # all-reduce is ordered after that lane only (run_program), the main lane
                    # is never stalled by the exchange
                    bi = self.dp.completes_bucket(l)
                    if bi is not None:
                        marks.append((prog.size(), bi))
                self.backward(mark)
            elif pipe_after > 0:
                # (experiment: the backbone's launches issued behind the first DISYOLO_PIPE_AFTER layers of the backward pass)
                done = [0]

                def late_start(_l):
                    done[0] += 1
                    if done[0] == pipe_after:
                        next_backbone()
                self.backward(late_start, sweep=True)
                if done[0] < pipe_after:
                    next_backbone()
            else:
                self.backward(sweep=True)
            bwd_end = prog.size()
            self._rec = None
            if self._overlap_rec:
                # what is left of the optimizer (the slices that became final last, gamma / beta, the finish) stays on the side
                # lane behind the last weight gradients; it needs the main lane's last batch-norm gradients
                # (data parallel: the remaining collectives go to the exchange lane, the sweeps wait for them here)
                L.lane_wait(L.lane_mark(0), 1)
                L.set_lane(1)
                self.optimizer_step(1.0 / self.dp.world_size if inl else 1.0)
                L.lane_mark_slot(1, self.SLOT_ALL)
                L.set_lane(0)
            else:
                self.optimizer_step(1.0 / self.dp.world_size if self.dp is not None else 1.0)
        return prog, marks, bwd_end

    def run_program(self) -> None:
        if self._progs is not None and self._pipe_in is not None and self.device.type == "cuda":
            q = self._parity
            try:
                self._run_pipelined_list(q)
            finally:
                self._ev_free[q].record(torch.cuda.current_stream())
                self._free_valid[q] = True
            return
        self._run_program()

    def _run_pipelined_list(self, q: int) -> None:
        cur = torch.cuda.current_stream()
        if cur == self.feed_stream:
            raise L.DisyoloError("train_step inside feed_context(): the step must run on the caller's stream, only set_batch "
                                 "(and what produces the batch) belongs on the feed stream")
        if self._fed_pending[q]:
            cur.wait_event(self._ev_fed[q])
            self._fed_pending[q] = False
        self._run_stream = cur
        self._run_program()

    def _run_program(self) -> None:
        if self._progs is not None:
            self._prog, self._prog_marks, self._bwd_end = self._progs[self._parity]
            self._parity ^= 1
        if self._graph is not None:
            self._graph.replay()
            return
        if self.dp is None or self.dp.inlist or os.environ.get("DISYOLO_DP_NOSEG") == "1":
            self._prog.run(join=not self._overlap)
            self._tail_open = self._overlap
            return
        self.dp.begin_step()
        # every recorded list owns its side lane: the bucket's all-reduce must be ordered after the
        # lane of the list that is running NOW (the pipelined step alternates between two lists)
        cuda = self.device.type == "cuda"          # (a plan-only net on the CPU drives the same cut loop in tests/test_dp_gloo.py)
        side = None
        if cuda:
            side = self._prog.side_stream(self.device)

        def on_lane(use_side: bool):
            if not cuda:
                return contextlib.nullcontext()
            return torch.cuda.stream(side if use_side else torch.cuda.current_stream())
        pos, first = 0, True
        for idx, what in self._prog_marks:
            self._prog.run(pos, idx, fork=first, join=False)
            first = False
            if isinstance(what, tuple):
                # SyncBN: the per-channel sums of one layer, on the lane that produced them
                _, t, lane = what
                with on_lane(lane == 1 and self.use_side_lane):
                    import torch.distributed as dist
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.dp.pg)
            else:
                with on_lane(self.use_side_lane):
                    self.dp.fire(what)
            pos = idx
        self._prog.run(pos, self._bwd_end, fork=first, join=True)
        self.dp.finish()
        self._prog.run(self._bwd_end, None, fork=False, join=False)

    def train_step(self, batch: Optional[Dict] = None, det_thresh: float = cfg.OBJ_THRESHOLD,
                   want_loss: bool = True):
        """``sess.run([net.total_loss, optimizer], feed_dict)`` (train_yolo3_mask.py:216)."""
        if not self.training:
            raise L.DisyoloError("train_step on a YOLONet built with training=False")
        if batch is not None:
            if self.pair:
                # (batch of this step, batch of the next one) before an even step; nothing before an odd one
                if self._parity_now() != 0 or len(batch) != 2:
                    raise L.DisyoloError("backbone_pair: pass the two batches of a step pair to the even step, None to the odd one")
                self.set_batch(batch[0], 0)
                self.set_batch(batch[1], 1)
            else:
                self.set_batch(batch)
        if self._prog is not None:
            self.run_program()
            return self.total_loss() if want_loss else None      # (total_loss joins an open tail)
        if self.pair:
            if self.dp is not None:
                raise L.DisyoloError("backbone_pair is a single-GPU option (no gradient exchange point in its step)")
            self.compute_losses(det_thresh, 1 if self._half == 0 else self._pair_P + 1)
            self.backward(sweep=True)
            self.optimizer_step()
            loss = self.total_loss() if want_loss else None
            self._use_half(1 - self._half)
            return loss
        self.compute_losses(det_thresh)
        if self.dp is not None and self.dp.inlist:
            self.backward(sweep=True)       # (every slice's collective is issued in front of its sweep)
            self.optimizer_step(1.0 / self.dp.world_size)
        elif self.dp is not None:
            self.dp.begin_step()
            self.backward(self.dp.on_layer_done)
            self.dp.finish()
            self.optimizer_step(1.0 / self.dp.world_size)
        else:
            self.backward(sweep=True)
            self.optimizer_step()
        # every term is the value for the weights BEFORE the update, like TF's fetch of total_loss
        # next to the train op (the l2 term comes out of the Adam sweep)
        return self.total_loss() if want_loss else None

    def _parity_now(self) -> int:
        """backbone_pair: 0 before an even step (which needs both batches set), 1 before an odd one"""
        return self._parity if self._progs is not None else self._half

    def summaries(self) -> Dict[str, float]:
        """the 7 tf.summary scalars (yolo/yolo3_net_pos.py:62,743-747,860)."""
        self.sync_lanes()
        v = self.losses.cpu().numpy()
        ml = float(self.mask_loss.cpu()[0])
        return {"object_loss": float(v[0]), "noobject_loss": float(v[1]), "class_loss": float(v[2]),
                "xy_loss": float(v[3]), "wh_loss": float(v[4]), "mask_loss": ml,
                "total_loss": float(self.total_loss().cpu())}
