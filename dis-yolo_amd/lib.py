"""ctypes binding of the gfx950 kernel library (``include/disyolo.h``).

There is no CPU fallback: if ``libdisyolo_hip.so`` is missing or an entry point
returns an error, this module raises.  PyTorch is used only to own device memory
and streams; kernels receive raw device pointers and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DISYOLO_LIB", os.path.join(_HERE, "libdisyolo_hip.so"))

GRAD_LD = 32
ROI_MAX = 16
ROI_W = 12
CONV_LEAKY, CONV_OUT_F32, CONV_STATS, CONV_BN_BWD_STATS, CONV_BN_FUSED, CONV_BN_BWD_FUSED = 1, 2, 4, 8, 16, 32


class DisyoloError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """mirror of ``disyolo_conv_desc`` (include/disyolo.h)"""
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("C0", C.c_int32), ("C1", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32),
        ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("in_div", C.c_int32), ("flags", C.c_int32),
        ("alpha", C.c_float), ("tile", C.c_int32),
        ("x0", C.c_void_p), ("x1", C.c_void_p), ("w", C.c_void_p),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("residual", C.c_void_p),
        ("y", C.c_void_p), ("stats", C.c_void_p),
        ("bn_x", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("bn_mean", C.c_void_p),
        ("bn_rstd", C.c_void_p), ("bn_partials", C.c_void_p), ("bn_alpha", C.c_float),
        ("bn_decay", C.c_float), ("bn_eps", C.c_float),
        ("y_act", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p),
        ("bn_moving_mean", C.c_void_p), ("bn_moving_var", C.c_void_p),
        ("bn_out_scale", C.c_void_p), ("bn_out_shift", C.c_void_p), ("bn_out_mean", C.c_void_p), ("bn_out_rstd", C.c_void_p),
        ("bn_dgamma", C.c_void_p), ("bn_dbeta", C.c_void_p), ("cluster_sync", C.c_void_p),
    ]


_SIGS = {
    "disyolo_version": (C.c_int, []),
    "disyolo_last_error": (C.c_char_p, []),
    "disyolo_conv_desc_size": (C.c_size_t, []),
    "disyolo_find_contours": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "disyolo_conv2d_stats_rows": (C.c_int, [C.POINTER(ConvDesc)]),
    "disyolo_conv2d_bn_bwd_stats_ok": (C.c_int, [C.POINTER(ConvDesc)]),
    "disyolo_conv2d_bn_fused_ok": (C.c_int, [C.POINTER(ConvDesc)]),
    "disyolo_cluster_sync_words": (C.c_int, [C.c_int]),
    "disyolo_cluster_sync_error": (C.c_int, [C.c_void_p, C.c_int]),
    "disyolo_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "disyolo_conv2d_tile": (C.c_int, [C.POINTER(ConvDesc)] + [C.POINTER(C.c_int)] * 4),
    "disyolo_conv_first_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "disyolo_conv12_fused_ok": (C.c_int, [C.c_int] * 3),
    "disyolo_conv12_fused_fwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 3 + [C.c_float, C.c_void_p]),
    "disyolo_block32_fused_ok": (C.c_int, [C.c_int] * 6),
    "disyolo_block64_fused_ok": (C.c_int, [C.c_int] * 4),
    "disyolo_pack_quad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "disyolo_dgrad_s2_quad_ok": (C.c_int, [C.c_int] * 5),
    "disyolo_dgrad_s2_quad": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    "disyolo_block64_fused_fwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "disyolo_block32_fused_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int] +
                                  [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_float, C.c_void_p]),
    "disyolo_conv2d_wgrad_workspace": (C.c_size_t, [C.POINTER(ConvDesc), C.c_int]),
    "disyolo_conv2d_wgrad": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_int, C.c_void_p]),
    "disyolo_conv2d_fp8_fwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                         C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "disyolo_conv_first_fwd_fp8": (C.c_int, [C.c_void_p] * 5 + [C.c_float] + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "disyolo_quant_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "disyolo_dequant_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "disyolo_pack_weights_fp8": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_float, C.c_void_p]),
    "disyolo_conv2d_wgrad_plan": (C.c_int, [C.POINTER(ConvDesc), C.c_int] + [C.POINTER(C.c_int)] * 4),
    "disyolo_conv_first_wgrad_workspace": (C.c_size_t, [C.c_int] * 4),
    "disyolo_conv_first_wgrad": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "disyolo_image_pad8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "disyolo_copy2d_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "disyolo_pack_weights": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "disyolo_pack_table_bytes": (C.c_size_t, [C.c_int]),
    "disyolo_pack_table_build": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "disyolo_pack_all": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "disyolo_bn_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64] + [C.c_void_p] * 4 +
                            [C.c_float, C.c_float] + [C.c_void_p] * 5),
    "disyolo_colstats_rows": (C.c_int, [C.c_int64, C.c_int]),
    "disyolo_colstats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "disyolo_bn_fold": (C.c_int, [C.c_void_p] * 4 + [C.c_float] + [C.c_void_p] * 2 + [C.c_int, C.c_void_p]),
    "disyolo_bn_act_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    "disyolo_bn_act_bwd_workspace": (C.c_size_t, [C.c_int64, C.c_int]),
    "disyolo_bn_act_bwd_partials_workspace": (C.c_size_t, [C.c_int]),
    "disyolo_bn_act_bwd_partials": (C.c_int, [C.c_void_p] * 9 + [C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "disyolo_bn_partial_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "disyolo_bn_finalize_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 4 + [C.c_float, C.c_float] +
                                 [C.c_void_p] * 5),
    "disyolo_bn_bwd_reduce_rows": (C.c_int, [C.c_int64, C.c_int]),
    "disyolo_bn_bwd_reduce": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "disyolo_bn_bwd_apply_sums": (C.c_int, [C.c_void_p] * 8 + [C.c_int64] + [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_float,
                                            C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "disyolo_bn_act_bwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                                       C.c_void_p]),
    "disyolo_upsample2x_bwd": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 7 + [C.c_void_p]),
    "disyolo_colsum_workspace": (C.c_size_t, [C.c_int64, C.c_int]),
    "disyolo_colsum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                 C.c_void_p]),
    "disyolo_detect_workspace": (C.c_size_t, [C.c_int] * 3),
    "disyolo_detect": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_float, C.c_float, C.c_int] +
                       [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p]),
    "disyolo_yolo_loss_workspace": (C.c_size_t, [C.c_int] * 3),
    "disyolo_yolo_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_float] +
                          [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]),
    "disyolo_shuffle_perm": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]),
    "disyolo_mask_rois": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 4 +
                          [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "disyolo_psroi_loss_workspace": (C.c_size_t, [C.c_int, C.c_int]),
    "disyolo_psroi_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 3 +
                           [C.c_float] + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p]),
    "disyolo_psroi_assemble": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 3),
    "disyolo_mask_paste": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "disyolo_letterbox": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "disyolo_confusion16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "disyolo_adam_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int64] + [C.c_float] * 5 + [C.c_int64, C.c_float,
                                                                                             C.c_void_p]),
    "disyolo_adam_step_dev": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int64] + [C.c_float] * 5 + [C.c_void_p, C.c_float,
                                                                                                 C.c_void_p]),
    "disyolo_adam_fused_workspace": (C.c_size_t, [C.c_int64]),
    "disyolo_adam_sweep_parts": (C.c_int, [C.c_int64]),
    "disyolo_adam_sweep": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int64, C.c_void_p] + [C.c_float] * 4 +
                           [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "disyolo_adam_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "disyolo_adam_finish_record": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]),
    "disyolo_adam_step_fused": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int64, C.c_void_p] + [C.c_float] * 4 +
                                [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "disyolo_add_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "disyolo_cmdlist_create": (C.c_void_p, []),
    "disyolo_cmdlist_destroy": (None, [C.c_void_p]),
    "disyolo_cmdlist_begin": (C.c_int, [C.c_void_p]),
    "disyolo_cmdlist_end": (C.c_int, []),
    "disyolo_cmdlist_size": (C.c_int, [C.c_void_p]),
    "disyolo_cmdlist_count": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "disyolo_cmdlist_set_lane": (C.c_int, [C.c_int]),
    "disyolo_cmdlist_sync": (C.c_int, [C.c_int, C.c_int]),
    "disyolo_cmdlist_mark": (C.c_int, [C.c_int]),
    "disyolo_cmdlist_wait": (C.c_int, [C.c_int, C.c_int]),
    "disyolo_cmdlist_mark_slot": (C.c_int, [C.c_int, C.c_int]),
    "disyolo_cmdlist_wait_slot": (C.c_int, [C.c_int, C.c_int]),
    "disyolo_cmdlist_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "disyolo_cmdlist_run_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "disyolo_cmdlist_side_stream": (C.c_void_p, [C.c_void_p]),
    "disyolo_cmdlist_lane_stream": (C.c_void_p, [C.c_void_p, C.c_int]),
    "disyolo_lanes_reserve": (C.c_int, [C.c_int]),
    "disyolo_lanes_report": (C.c_int, [C.c_char_p, C.c_int]),
    "disyolo_comm_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "disyolo_comm_unique_id": (C.c_int, [C.c_void_p]),
    "disyolo_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "disyolo_comm_destroy": (C.c_int, [C.c_void_p]),
    "disyolo_comm_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "disyolo_comm_reduce_scatter_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "disyolo_comm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "disyolo_cast_f32_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "disyolo_cast_bf16_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "disyolo_polygon_mask": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p]),
    "disyolo_aug_place": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "disyolo_aug_place_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "disyolo_aug_salt_pepper": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "disyolo_aug_change_light": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "disyolo_aug_motion_blur3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "disyolo_aug_to_float": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "disyolo_crc32c": (C.c_uint32, [C.c_void_p, C.c_size_t, C.c_uint32]),
    "disyolo_l2_workspace": (C.c_size_t, [C.c_int64]),
    "disyolo_l2_loss": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def load() -> C.CDLL:
    """Load the kernel library; raises DisyoloError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DisyoloError(
                "%s not found: build it with `make -C dis-yolo_amd/csrc` (or __graft_entry__.build()); "
                "there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.disyolo_conv_desc_size() != C.sizeof(ConvDesc):
            raise DisyoloError("%s was built from a different include/disyolo.h (conv descriptor %d bytes, this "
                               "binding %d): rebuild it" % (LIB_PATH, lib.disyolo_conv_desc_size(), C.sizeof(ConvDesc)))
        _lib = lib
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise DisyoloError("%s failed (%d): %s" % (what, rc, load().disyolo_last_error().decode()))


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need(t: torch.Tensor, dtype, name: str) -> None:
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise DisyoloError("%s must be a contiguous CUDA tensor of %s (got %s on %s)" % (name, dtype, t.dtype, t.device))


def _need_shape(t: torch.Tensor, shape, name: str) -> None:
    if tuple(t.shape) != tuple(shape):
        raise DisyoloError("%s must have shape %s (got %s)" % (name, tuple(shape), tuple(t.shape)))


class Workspace:
    """Grow-only device scratch buffer handed to the kernels that need one.  Once a command
    list has captured its address it is frozen: a later request that does not fit raises
    instead of silently moving the buffer."""

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(1 << 20, dtype=torch.uint8, device=device)
        self.frozen = False

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf.numel() < nbytes:
            if self.frozen:
                raise DisyoloError("workspace of %d bytes is frozen by a recorded command list; %d requested"
                                   % (self.buf.numel(), nbytes))
            self.buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self.buf


class CmdList:
    """Recorded sequence of kernel-library calls replayed by one C call (csrc/runtime.hip)."""

    def __init__(self):
        self.h = load().disyolo_cmdlist_create()
        self.keep = []           # tensors / descriptors referenced by the recorded commands
        self._side = None        # side_stream()'s torch wrapper
        self._lanes = {}

    def __del__(self):
        try:
            if self.h:
                if torch.cuda.is_available() and (self._side is not None or self._lanes):
                    # a side lane may still be running: the caller's stream waits before streams / buffers are released
                    for st in [self._side] + list(self._lanes.values()):
                        if st is not None:
                            torch.cuda.current_stream(st.device).wait_stream(st)
                load().disyolo_cmdlist_destroy(self.h)
        except Exception:
            pass

    def __enter__(self):
        global TIMER
        if TIMER is not None:
            raise DisyoloError("cannot record a command list while the kernel timer is active")
        _check(load().disyolo_cmdlist_begin(self.h), "cmdlist_begin")
        global CURRENT_LANE
        CURRENT_LANE = 0
        return self

    def __exit__(self, *exc):
        _check(load().disyolo_cmdlist_end(), "cmdlist_end")
        global CURRENT_LANE
        CURRENT_LANE = 0
        return False

    def size(self) -> int:
        return load().disyolo_cmdlist_size(self.h)

    def count(self, what: str, lane: int) -> int:
        """packets a replay puts on ``lane``: what = "launches" | "records" | "waits" """
        return load().disyolo_cmdlist_count(self.h, {"launches": 0, "records": 1, "waits": 2}[what], lane)

    def run(self, first: int = 0, last: Optional[int] = None, fork: bool = True, join: bool = True) -> None:
        last = self.size() if last is None else last
        _check(load().disyolo_cmdlist_run_ex(self.h, first, last, _stream(), (1 if fork else 0) | (2 if join else 0)),
               "cmdlist_run")

    def side_stream(self, device) -> "torch.cuda.Stream":
        """the side lane as a torch stream (to order an RCCL collective, or the caller's stream, after it).  The wrapper
        lives and dies with this list: the HIP stream is destroyed with it"""
        if self._side is None:
            self._side = torch.cuda.ExternalStream(load().disyolo_cmdlist_side_stream(self.h), device=device)
        return self._side

    def lane_stream(self, lane: int, device) -> "torch.cuda.Stream":
        """lane 1..3 of this list as a torch stream (same ownership as side_stream)"""
        if lane == 1:
            return self.side_stream(device)
        if lane not in self._lanes:
            self._lanes[lane] = torch.cuda.ExternalStream(load().disyolo_cmdlist_lane_stream(self.h, lane), device=device)
        return self._lanes[lane]


def same_pads(size: int, k: int, s: int):
    """TF 'SAME' geometry: (out, pad_before)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


def make_conv_desc(x0, w_packed, y, ksize, stride, *, x1=None, scale=None, shift=None, residual=None, stats=None,
                   leaky=False, out_f32=False, alpha=0.1, tile=0, in_div=1, pads=None, out_hw=None,
                   bn_bwd=None, bn_fused=None, bn_bwd_fused=None) -> ConvDesc:
    """``bn_bwd`` = (x, scale, shift, mean, rstd, partials, alpha) of the batch-normalised layer whose output
    gradient ``y`` becomes final with this conv: its batch-norm backward sums are emitted by the epilogue (only the
    3x3 patch kernel can: check ``conv2d_bn_bwd_stats_ok`` on the descriptor without it first).
    ``bn_fused`` = dict(y_act, gamma, beta, mm, mv, scale, shift, mean, rstd, decay, eps, sync): training-mode batch norm
    inside the launch (DISYOLO_CONV_BN_FUSED; ``stats`` must be given, ``y`` receives the raw conv output, ``y_act`` the
    activation; check ``conv2d_bn_fused_ok`` on a descriptor without it first).
    ``bn_bwd_fused`` = dict(dgamma, dbeta, sync) on top of ``bn_bwd``: the target's whole batch-norm backward inside this
    data-gradient conv (``y`` receives dx of the target's conv output)."""
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    if out_hw is None:
        Ho, pt = same_pads(H, ksize, stride)
        Wo, pl = same_pads(W, ksize, stride)
    else:
        Ho, Wo = out_hw
        pt, pl = pads
    if pads is not None:
        pt, pl = pads
    d = ConvDesc()
    d.B, d.H, d.W, d.C0, d.C1 = B, H, W, C0, C1
    d.Ho, d.Wo, d.Cout = Ho, Wo, y.shape[-1]
    d.ksize, d.stride, d.pad_t, d.pad_l, d.in_div = ksize, stride, pt, pl, in_div
    d.flags = (CONV_LEAKY if leaky else 0) | (CONV_OUT_F32 if out_f32 else 0) | (CONV_STATS if stats is not None else 0)
    d.alpha, d.tile = alpha, tile
    if tile == 0 and TUNED:
        d.tile = TUNED.get(conv_shape_key(d), 0)
    d.x0, d.x1, d.w = _p(x0), _p(x1), _p(w_packed)
    d.scale, d.shift, d.residual = _p(scale), _p(shift), _p(residual)
    d.y, d.stats = _p(y), _p(stats)
    if bn_bwd is not None:
        d.flags |= CONV_BN_BWD_STATS
        d.bn_x, d.bn_scale, d.bn_shift, d.bn_mean, d.bn_rstd, d.bn_partials = (_p(t) for t in bn_bwd[:6])
        d.bn_alpha = bn_bwd[6]
    if bn_fused is not None:
        f = bn_fused
        if stats is None or leaky or scale is not None or shift is not None or residual is not None:
            raise DisyoloError("make_conv_desc(bn_fused=...): needs stats, and no scale / shift / leaky / residual")
        d.flags |= CONV_BN_FUSED
        d.y_act, d.bn_gamma, d.bn_beta = _p(f["y_act"]), _p(f["gamma"]), _p(f["beta"])
        d.bn_moving_mean, d.bn_moving_var = _p(f.get("mm")), _p(f.get("mv"))
        d.bn_out_scale, d.bn_out_shift, d.bn_out_mean, d.bn_out_rstd = _p(f["scale"]), _p(f["shift"]), _p(f["mean"]), _p(f["rstd"])
        d.bn_decay, d.bn_eps = f["decay"], f["eps"]
        d.cluster_sync = _p(f["sync"])
    if bn_bwd_fused is not None:
        f = bn_bwd_fused
        if bn_bwd is None:
            raise DisyoloError("make_conv_desc(bn_bwd_fused=...): needs bn_bwd")
        d.flags |= CONV_BN_BWD_FUSED
        d.bn_dgamma, d.bn_dbeta, d.cluster_sync = _p(f["dgamma"]), _p(f["dbeta"]), _p(f["sync"])
    # the struct only holds raw pointers: keep the tensors alive as long as the descriptor
    d._keepalive = (x0, x1, w_packed, scale, shift, residual, y, stats, bn_bwd, bn_fused, bn_bwd_fused)
    return d


def conv2d_bn_fused_ok(d: ConvDesc) -> bool:
    """can this descriptor (its shape and tile) run batch norm inside the launch -- a kernel with the epilogue and a grid
    that is resident at once on this device?  (make_conv_desc(bn_fused=...) / (bn_bwd_fused=...))"""
    return load().disyolo_conv2d_bn_fused_ok(C.byref(d)) == 1


def cluster_sync_buffer(cout: int, device) -> torch.Tensor:
    """the counters of one layer's in-launch exchange (zeroed once; the launches leave them zero) + its error word"""
    return torch.zeros(load().disyolo_cluster_sync_words(cout), dtype=torch.int32, device=device)


def cluster_sync_error(buf: torch.Tensor, cout: int) -> int:
    return load().disyolo_cluster_sync_error(_p(buf), cout)


def conv2d_bn_bwd_stats_ok(d: ConvDesc) -> bool:
    """does this descriptor run a kernel that can emit the batch-norm backward sums (make_conv_desc(bn_bwd=...))?"""
    return load().disyolo_conv2d_bn_bwd_stats_ok(C.byref(d)) == 1


def conv2d_stats_rows(d: ConvDesc) -> int:
    r = load().disyolo_conv2d_stats_rows(C.byref(d))
    if r < 0:
        raise DisyoloError("conv2d_stats_rows: bad descriptor")
    return r


def conv_shape_key(d: ConvDesc):
    """what the tile choice depends on: the GEMM shape and the gather variant (not the epilogue:
    the batch-norm partial-sum rows are sized from a descriptor without one)"""
    return (d.B, d.H, d.W, d.C0, d.C1, d.Ho, d.Wo, d.Cout, d.ksize, d.stride, d.in_div)


# shape key -> tile code, filled by ConvTuner (YOLONet.autotune); consulted by make_conv_desc
TUNED: dict = {}
TUNE_CANDIDATES = (3, 0x203, 6, 0x206, 2, 0x202, 0x204, 10, 12, 0x20c, 0x10c, 0x108, 0x20d, 16, 17, 18, 19, 20, 21, 24, 25)
# (tiles 26 / 27 / 28 -- two-wave 32x64, 64x32, 32x128 GEMM tiles, round 5 -- cover every shape and won nowhere in the step:
# profiles/r05_two_wave_tiles.txt; pass them explicitly to try them)


class ConvTuner:
    """In-sequence tile autotuner.  Stand-alone timing loops keep a layer's operands hot in L2
    and rank the tiles differently from how they behave inside the step (measured: DESIGN.md),
    so the candidates are timed where they run: while ``active`` every conv2d_fwd launch is
    bracketed with HIP events and uses the candidate of the current pass; the best candidate
    per shape key goes to ``TUNED``."""

    def __init__(self, candidates=TUNE_CANDIDATES):
        self.candidates = tuple(candidates)
        self.current = 0              # tile code of the running pass (0 = launcher heuristic)
        self.events = {}              # (key, cand) -> [(start, end)]
        self.stats_rows = {}          # stats buffer address -> partial-sum rows the last launch into it wrote
        self.covers = {}              # (key, cand) -> does the candidate's kernel cover the shape (else the launcher falls back)

    def launch(self, d: ConvDesc) -> None:
        key = conv_shape_key(d)
        keep = d.tile
        d.tile = self.current
        ck = (key, self.current)
        if ck not in self.covers:
            # a tile that does not cover the shape resolves to the heuristic's pick: timing it would enter the heuristic into
            # the table under a foreign id whenever it wins by noise
            self.covers[ck] = (not self.current) or conv2d_tile(d)[0] == (self.current & 0xff)
        if d.stats:
            # the number of batch-norm partial-sum rows depends on the tile: the finalize that follows must sum
            # exactly the rows THIS candidate writes, or the tuning passes run on garbage statistics
            self.stats_rows[d.stats] = conv2d_stats_rows(d)
        # a candidate tile may have no in-launch batch-norm epilogue, or a grid that is not resident at once: the tuning
        # passes run every conv without it (the caller issues the separate batch-norm launches while a tuner is active)
        keep_flags = d.flags
        d.flags &= ~(CONV_BN_FUSED | CONV_BN_BWD_FUSED)
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = load().disyolo_conv2d_fwd(C.byref(d), _stream())
        e.record()
        d.tile = keep
        d.flags = keep_flags
        _check(rc, "conv2d_fwd")
        if self.covers[ck]:
            self.events.setdefault(ck, []).append((s, e))

    def table(self):
        """{key: {cand: median ms}}"""
        torch.cuda.synchronize()
        out = {}
        for (key, cand), evs in self.events.items():
            ts = sorted(s.elapsed_time(e) for s, e in evs)
            out.setdefault(key, {})[cand] = ts[len(ts) // 2]
        return out

    def commit(self, min_gain: float = 0.02) -> dict:
        """pick the fastest candidate per key; keep the heuristic unless beaten by min_gain"""
        picks = {}
        for key, row in self.table().items():
            best = min(row, key=row.get)
            base = row.get(0)
            if base is not None and row[best] > base * (1.0 - min_gain):
                best = 0
            picks[key] = best
            if best:
                TUNED[key] = best
            else:
                TUNED.pop(key, None)
        return picks


TUNER: Optional[ConvTuner] = None


class KernelTimer:
    """Brackets selected launches with HIP events on the launch stream (bench.py roofline)."""

    def __init__(self):
        self.records = {}   # name -> [flops_total, [(start, end), ...]]

    def run(self, name: str, flops: float, fn, nbytes: float = 0.0) -> None:
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        rec = self.records.setdefault(name, [0.0, [], 0.0])
        rec[0] += flops
        rec[1].append((s, e))
        rec[2] += nbytes

    def summary(self):
        out = {}
        for name, (flops, evs, nbytes) in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e in evs)
            out[name] = {"launches": len(evs), "ms_total": ms, "flops_total": flops, "bytes_total": nbytes}
        return out


TIMER: Optional[KernelTimer] = None


def conv_flops(d: ConvDesc) -> float:
    """algorithmic FLOPs of the convolution this descriptor belongs to: 2*M*Cout*K; for the
    transposed (data-gradient) gather only 1/in_div^2 of the taps are real work"""
    K = d.ksize * d.ksize * (d.C0 + d.C1)
    return 2.0 * d.B * d.Ho * d.Wo * d.Cout * K / float(d.in_div * d.in_div)


def conv_bytes(d: ConvDesc) -> float:
    """algorithmic HBM bytes of one launch: input(s) + packed weights + output (+ the residual it adds), each once"""
    K = d.ksize * d.ksize * (d.C0 + d.C1)
    out_b = d.B * d.Ho * d.Wo * d.Cout * (4 if d.flags & CONV_OUT_F32 else 2)
    return float(d.B * d.H * d.W * d.C0 * 2 + d.B * (d.H // 2) * (d.W // 2) * d.C1 * 2 + d.Cout * K * 2 + out_b
                 + (d.B * d.Ho * d.Wo * d.Cout * 2 if d.residual else 0))


_WAVES = {1: (2, 2), 2: (2, 2), 3: (2, 2), 4: (4, 1), 5: (4, 1), 6: (2, 2), 7: (4, 1), 8: (4, 2), 9: (2, 2),
          10: (2, 2), 11: (2, 2), 12: (4, 2), 13: (2, 2), 14: (2, 2), 15: (4, 2), 26: (1, 2), 27: (2, 1), 28: (1, 2), 29: (4, 2)}


def conv2d_tile(d: ConvDesc):
    """(tile id, BM, BN, BK, stages) of the conv_igemm_kernel instance the launcher will pick"""
    bm, bn, bk, st = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    tid = load().disyolo_conv2d_tile(C.byref(d), C.byref(bm), C.byref(bn), C.byref(bk), C.byref(st))
    if tid < 0:
        raise DisyoloError("conv2d_tile: bad descriptor")
    return tid, bm.value, bn.value, bk.value, st.value


def conv2d_fwd(d: ConvDesc) -> None:
    if TUNER is not None:
        TUNER.launch(d)
        return
    if TIMER is not None:
        if not hasattr(d, "_tname"):
            tid, bm, bn, bk, st = conv2d_tile(d)
            if tid in (24, 25):
                d._tname = "conv_flat_kernel<%s,2,%d>" % ("2,3,2,1" if tid == 24 else "4,3,1,2", st)
            elif tid == 21:
                d._tname = "conv1x1_stream_kernel<%d>" % (bn // 16)
            elif tid == 20:
                d._tname = "conv_stream_kernel<8,3,4>"
            elif tid >= 16:
                d._tname = "conv_halo_kernel<%d,3,%d>" % (4 if tid == 17 else 8, 2 if tid == 18 else 1 if tid == 19 else 4)
            else:
                # (the last argument: 1 = the instance with the fused batch-norm backward epilogue, round 6)
                d._tname = "conv_igemm_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d>" % ((bm, bn) + _WAVES.get(tid, (0, 0)) +
                                                                               (bk, st, d.ksize, 2 if 13 <= tid <= 15 else 1,
                                                                                1 if d.flags & CONV_BN_BWD_FUSED else 0))
        TIMER.run(d._tname, conv_flops(d), lambda: _check(load().disyolo_conv2d_fwd(C.byref(d), _stream()),
                                                           "conv2d_fwd"), conv_bytes(d))
        return
    _check(load().disyolo_conv2d_fwd(C.byref(d), _stream()), "conv2d_fwd")


def conv2d_fp8_fwd(d: ConvDesc, w8, escale, eshift, y8, out_scale: float, y16=None, residual8=None,
                   residual_scale: float = 0.0) -> None:
    """fp8 forward conv: d.x0 = e4m3 input; outputs e4m3 (y8, of y / out_scale) and/or bf16 (y16)"""
    if TIMER is not None:
        name = "conv_fp8_kernel<%d,%d,3>" % (128 if d.Cout > 64 else (64 if d.Cout > 32 else 32), 64 if d.C0 % 64 == 0 else 32)
        TIMER.run(name, conv_flops(d), lambda: _check(load().disyolo_conv2d_fp8_fwd(
            C.byref(d), _p(w8), _p(escale), _p(eshift), _p(residual8), residual_scale, _p(y8), out_scale, _p(y16), _stream()),
            "conv2d_fp8_fwd"))
        return
    _check(load().disyolo_conv2d_fp8_fwd(C.byref(d), _p(w8), _p(escale), _p(eshift), _p(residual8), residual_scale, _p(y8),
                                         out_scale, _p(y16), _stream()), "conv2d_fp8_fwd")


def conv_first_fwd_fp8(images, w_hwio, scale, shift, y8, out_scale: float, alpha=0.1) -> None:
    _need(images, torch.float32, "images")
    _need(y8, torch.uint8, "y8")
    B, H, W, _ = images.shape
    _check(load().disyolo_conv_first_fwd_fp8(_p(images), _p(w_hwio), _p(scale), _p(shift), _p(y8), out_scale, B, H, W,
                                             y8.shape[-1], alpha, _stream()), "conv_first_fwd_fp8")


def quant_fp8(x, y8, scale: float) -> None:
    """y8 (uint8 holding OCP e4m3) = e4m3(x / scale); x bf16 or f32"""
    if x.dtype not in (torch.bfloat16, torch.float32) or not x.is_contiguous():
        raise DisyoloError("quant_fp8: x must be contiguous bf16 or f32")
    _check(load().disyolo_quant_fp8(_p(x), int(x.dtype == torch.float32), _p(y8), x.numel(), scale, _stream()), "quant_fp8")


def dequant_fp8(x8, y, scale: float) -> None:
    _need(y, torch.float32, "y")
    _check(load().disyolo_dequant_fp8(_p(x8), _p(y), x8.numel(), scale, _stream()), "dequant_fp8")


def pack_weights_fp8(w_hwio, w8, ksize, cin, cout, scale: float) -> None:
    _check(load().disyolo_pack_weights_fp8(_p(w_hwio), _p(w8), ksize, cin, cout, scale, _stream()), "pack_weights_fp8")


def conv12_fused_ok(B: int, H: int, W: int) -> bool:
    return load().disyolo_conv12_fused_ok(B, H, W) == 1


def conv12_fused_fwd(images, w1_hwio, scale1, shift1, w2_packed, scale2, shift2, y, alpha=0.1) -> None:
    """conv1 + conv2 (both in inference mode) in one launch; conv1's output is not materialised"""
    _need(images, torch.float32, "images")
    _need(y, torch.bfloat16, "y")
    B, H, W, _ = images.shape
    _need_shape(y, (B, H // 2, W // 2, 64), "conv12_fused_fwd: y")
    fn = lambda: _check(load().disyolo_conv12_fused_fwd(_p(images), _p(w1_hwio), _p(scale1), _p(shift1), _p(w2_packed), _p(scale2),
                                                        _p(shift2), _p(y), B, H, W, alpha, _stream()), "conv12_fused_fwd")
    if TIMER is not None:
        flops = 2.0 * B * H * W * 32 * 27 + 2.0 * B * (H // 2) * (W // 2) * 64 * 288
        TIMER.run("conv12_fused_kernel", flops, fn, float(B * H * W * 12 + B * (H // 2) * (W // 2) * 128))
        return
    fn()


def block32_fused_ok(B: int, H: int, W: int, C0: int, C1: int, post: int) -> bool:
    return load().disyolo_block32_fused_ok(B, H, W, C0, C1, post) == 1


def block32_fused_fwd(x0, x1, wA, scaleA, shiftA, wB, scaleB, shiftB, y, post=0, wC=None, biasC=None, alpha=0.1) -> None:
    """[1x1 (C0 + up2(C1)) -> 32] -> [3x3 32 -> 64] with folded batch norms in one launch; post 0: + residual x0 -> bf16 y
    (the first residual block), post 1: -> [1x1 64 -> 9] + bias -> f32 y (the mask head).  The 32- and 64-channel
    intermediates are not materialised."""
    _need(x0, torch.bfloat16, "x0")
    _need(y, torch.bfloat16 if post == 0 else torch.float32, "y")
    B, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[-1]
    # the C launcher takes B, H, W from x0 and cannot see the other buffers' sizes
    if x1 is not None:
        _need(x1, torch.bfloat16, "x1")
        _need_shape(x1, (B, H // 2, W // 2, C1), "block32_fused_fwd: x1")
    _need_shape(y, (B, H, W, 9 if post else 64), "block32_fused_fwd: y")
    fn = lambda: _check(load().disyolo_block32_fused_fwd(_p(x0), _p(x1), C0, C1, _p(wA), _p(scaleA), _p(shiftA), _p(wB), _p(scaleB),
                                                         _p(shiftB), post, _p(wC), _p(biasC), _p(y), B, H, W, alpha, _stream()),
                        "block32_fused_fwd")
    if TIMER is not None:
        M = B * H * W
        flops = 2.0 * M * (32 * (C0 + C1) + 64 * 288 + (64 * 9 if post else 0))
        nbytes = M * C0 * 2 + (M // 4) * C1 * 2 + (M * 36 if post else M * 128)
        TIMER.run("block32_kernel<%d,%d>" % (C0 + C1, post), flops, fn, float(nbytes))
        return
    fn()


def dgrad_s2_quad_ok(B: int, Hdy: int, Wdy: int, Cdy: int, C: int) -> bool:
    return load().disyolo_dgrad_s2_quad_ok(B, Hdy, Wdy, Cdy, C) == 1


def pack_quad(w_hwio, wq) -> None:
    """f32 HWIO master [3,3,C,Cdy] of a stride-2 conv -> the bf16 operand [4C, 9*Cdy] of dgrad_s2_quad"""
    _need(w_hwio, torch.float32, "w")
    _need(wq, torch.bfloat16, "wq")
    C_, Cdy = w_hwio.shape[2], w_hwio.shape[3]
    _check(load().disyolo_pack_quad(_p(w_hwio), _p(wq), C_, Cdy, _stream()), "pack_quad")


def dgrad_s2_quad(dy, wq, dx, accumulate: bool = False) -> None:
    """data gradient of a 3x3 stride-2 conv (even input size, C <= 64): dx [B,2H,2W,C] (+)= quad conv of dy [B,H,W,Cdy]"""
    _need(dy, torch.bfloat16, "dy")
    _need(dx, torch.bfloat16, "dx")
    B, H, W, Cdy = dy.shape
    C_ = dx.shape[-1]
    fn = lambda: _check(load().disyolo_dgrad_s2_quad(_p(dy), _p(wq), _p(dx), _p(dx) if accumulate else None, B, H, W, Cdy, C_,
                                                      _stream()), "dgrad_s2_quad")
    if TIMER is not None:
        TIMER.run("conv_igemm_kernel<192,128,4,2,64,2,3,1>", 2.0 * B * H * W * 4 * C_ * 4 * Cdy, fn,
                  float(B * H * W * (Cdy + 4 * C_) * 2))
        return
    fn()


def block64_fused_ok(B: int, H: int, W: int, C0: int) -> bool:
    return load().disyolo_block64_fused_ok(B, H, W, C0) == 1


def block64_fused_fwd(x, wA, scaleA, shiftA, wB, scaleB, shiftB, y, alpha=0.1) -> None:
    """a residual block of the 144^2 maps, [1x1 128 -> 64] -> [3x3 64 -> 128] + x with folded batch norms, in one launch"""
    _need(x, torch.bfloat16, "x")
    _need(y, torch.bfloat16, "y")
    B, H, W, C0 = x.shape
    _need_shape(y, (B, H, W, C0), "block64_fused_fwd: y")
    fn = lambda: _check(load().disyolo_block64_fused_fwd(_p(x), _p(wA), _p(scaleA), _p(shiftA), _p(wB), _p(scaleB), _p(shiftB), _p(y),
                                                         B, H, W, C0, alpha, _stream()), "block64_fused_fwd")
    if TIMER is not None:
        M = B * H * W
        TIMER.run("block64_kernel", 2.0 * M * (64 * 128 + 128 * 576), fn, float(M * 512))
        return
    fn()


def conv_first_fwd(images, w_hwio, scale, shift, y, alpha=0.1) -> None:
    _need(images, torch.float32, "images")
    _need(y, torch.bfloat16, "y")
    B, H, W, _ = images.shape
    _check(load().disyolo_conv_first_fwd(_p(images), _p(w_hwio), _p(scale), _p(shift), _p(y), B, H, W, y.shape[-1],
                                         alpha, _stream()), "conv_first_fwd")


WGRAD_IM2COL, WGRAD_PARTIAL_ONLY, WGRAD_REDUCE_ONLY = 1, 2, 4


def wgrad_stages(n: int) -> int:
    """opts bits for the im2col weight-gradient kernel's pipeline depth (2..4 stages)"""
    if not 2 <= int(n) <= 4:
        raise DisyoloError("wgrad_stages: 2..4 pipeline stages (got %r)" % (n,))
    return (int(n) - 1) << 4


def conv2d_wgrad_workspace(d: ConvDesc, opts: int = 0) -> int:
    return load().disyolo_conv2d_wgrad_workspace(C.byref(d), opts)


def conv2d_wgrad(d: ConvDesc, dy, dy_ld: int, dw, ws: Workspace, opts: int = 0) -> None:
    """dw = weight gradient of the conv ``d`` describes.  ``d.tile`` is not read (it belongs to conv2d_fwd);
    ``opts`` = WGRAD_* tuning / timing switches, 0 in the product path."""
    need = conv2d_wgrad_workspace(d, opts)
    buf = ws.get(need)

    def call(o):
        _check(load().disyolo_conv2d_wgrad(C.byref(d), _p(dy), dy_ld, _p(dw), _p(buf), buf.numel(), o, _stream()),
               "conv2d_wgrad")
    if TIMER is not None and not (opts & (WGRAD_PARTIAL_ONLY | WGRAD_REDUCE_ONLY)):
        # the partial-sum kernel and the slab reduction as two timed launches (same work, same order)
        kind, tn, ring, splits = conv2d_wgrad_plan(d, opts)
        waves = 8 if os.environ.get("DISYOLO_WG3_WAVES") == "8" else 4
        name = ("conv_wgrad3x3_kernel<%d,%d,3,%d,0>" % (tn, ring, waves)) if kind == 1 else ("conv_wgrad_kernel<%d,3>" % tn)
        if kind == 1 and d.stride == 2:
            name = "conv_wgrad3x3_kernel<64,8,3,4,1>"
        TIMER.run(name, conv_flops(d), lambda: call(opts | WGRAD_PARTIAL_ONLY))
        if splits > 1:
            TIMER.run("slab_reduce_kernel", 0.0, lambda: call(opts | WGRAD_REDUCE_ONLY))
        return
    call(opts)


def conv2d_wgrad_plan(d: ConvDesc, opts: int = 0):
    """(kind, channel tile, ring slots, pixel splits) of the weight-gradient launch for this descriptor"""
    v = [C.c_int(0) for _ in range(4)]
    _check(load().disyolo_conv2d_wgrad_plan(C.byref(d), opts, *[C.byref(x) for x in v]), "conv2d_wgrad_plan")
    return tuple(x.value for x in v)


def conv_first_wgrad(images, dy, dw, ws: Workspace) -> None:
    B, H, W, _ = images.shape
    cout = dy.shape[-1]
    need = load().disyolo_conv_first_wgrad_workspace(B, H, W, cout)
    buf = ws.get(need)
    _check(load().disyolo_conv_first_wgrad(_p(images), _p(dy), _p(dw), B, H, W, cout, _p(buf), buf.numel(), _stream()),
           "conv_first_wgrad")


def image_pad8(images, out) -> None:
    _check(load().disyolo_image_pad8(_p(images), _p(out), images.numel() // 3, _stream()), "image_pad8")


def copy2d_f32(src, dst, rows, cols, src_ld, dst_ld) -> None:
    _check(load().disyolo_copy2d_f32(_p(src), _p(dst), rows, cols, src_ld, dst_ld, _stream()), "copy2d_f32")


def pack_weights(w_hwio, w_fwd, w_dgrad, ksize, cin, cout, cout_pad=0) -> None:
    _check(load().disyolo_pack_weights(_p(w_hwio), _p(w_fwd), _p(w_dgrad), ksize, cin, cout, max(cout_pad, cout),
                                       _stream()), "pack_weights")


class PackJob(C.Structure):
    """mirror of ``disyolo_pack_job``"""
    _fields_ = [("w_hwio", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p),
                ("ksize", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32), ("cout_pad", C.c_int32)]


class PackTable:
    """Device-resident job table for pack_all: every trainable layer re-packed by one launch."""

    def __init__(self, jobs, device):
        n = len(jobs)
        arr = (PackJob * n)()
        for i, (w, wf, wd, k, cin, cout, pad) in enumerate(jobs):
            arr[i] = PackJob(_p(w), _p(wf), _p(wd), k, cin, cout, pad)
        nbytes = load().disyolo_pack_table_bytes(n)
        host = (C.c_char * nbytes)()
        blocks = C.c_int(0)
        _check(load().disyolo_pack_table_build(arr, n, host, C.byref(blocks)), "pack_table_build")
        self.n, self.blocks = n, blocks.value
        self.dev = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(device)
        self.keep = jobs

    def run(self) -> None:
        _check(load().disyolo_pack_all(_p(self.dev), self.n, self.blocks, _stream()), "pack_all")


def bn_finalize(stats, rows, C_, count, gamma, beta, mm, mv, decay, eps, scale, shift, mean, rstd) -> None:
    _check(load().disyolo_bn_finalize(_p(stats), rows, C_, count, _p(gamma), _p(beta), _p(mm), _p(mv), decay, eps,
                                      _p(scale), _p(shift), _p(mean), _p(rstd), _stream()), "bn_finalize")


def colstats_rows(rows, C_) -> int:
    r = load().disyolo_colstats_rows(rows, C_)
    if r < 0:
        raise DisyoloError("colstats_rows: bad shape")
    return r


def colstats(x, stats, rows, C_) -> None:
    _check(load().disyolo_colstats(_p(x), _p(stats), rows, C_, _stream()), "colstats")


def bn_act_bwd_partials(dy, x, scale, shift, mean, rstd, dx, dgamma, dbeta, rows, C_, partials, part_rows,
                        ws: Workspace, alpha=0.1, shortcut_grad=None, shortcut_accumulate=False) -> None:
    """bn_act_bwd whose column reduction was done by the conv that produced dy (make_conv_desc(bn_bwd=...))"""
    _need(partials, torch.float32, "partials")
    if partials.numel() < part_rows * C_ * 2:
        raise DisyoloError("bn_act_bwd_partials: partials smaller than part_rows x C x 2")
    buf = ws.get(load().disyolo_bn_act_bwd_partials_workspace(C_))
    _check(load().disyolo_bn_act_bwd_partials(_p(dy), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), _p(dx), _p(dgamma),
                                              _p(dbeta), rows, C_, alpha, _p(partials), part_rows, _p(shortcut_grad),
                                              int(shortcut_accumulate), _p(buf), buf.numel(), _stream()), "bn_act_bwd_partials")


def bn_partial_sums(partials, rows, C_, sums) -> None:
    """SyncBN: conv-epilogue partial rows -> this rank's f64 sums [C,2]"""
    _need(sums, torch.float64, "sums")
    _check(load().disyolo_bn_partial_sums(_p(partials), rows, C_, _p(sums), _stream()), "bn_partial_sums")


def bn_finalize_sums(sums, C_, count, gamma, beta, mm, mv, decay, eps, scale, shift, mean, rstd) -> None:
    _need(sums, torch.float64, "sums")
    _check(load().disyolo_bn_finalize_sums(_p(sums), C_, count, _p(gamma), _p(beta), _p(mm), _p(mv), decay, eps, _p(scale),
                                           _p(shift), _p(mean), _p(rstd), _stream()), "bn_finalize_sums")


def bn_bwd_reduce(dy, x, scale, shift, mean, rstd, rows, C_, sums, ws: Workspace, alpha=0.1) -> None:
    _need(sums, torch.float64, "sums")
    buf = ws.get(load().disyolo_bn_bwd_reduce_rows(rows, C_) * C_ * 2 * 4)
    _check(load().disyolo_bn_bwd_reduce(_p(dy), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), rows, C_, alpha, _p(sums),
                                        _p(buf), buf.numel(), _stream()), "bn_bwd_reduce")


def bn_bwd_apply_sums(dy, x, scale, shift, mean, rstd, local_sums, global_sums, count, dx, dgamma, dbeta, rows, C_,
                      ws: Workspace, alpha=0.1, shortcut_grad=None, shortcut_accumulate=False) -> None:
    buf = ws.get(2 * C_ * 4)
    _check(load().disyolo_bn_bwd_apply_sums(_p(dy), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), _p(local_sums),
                                            _p(global_sums), count, _p(dx), _p(dgamma), _p(dbeta), rows, C_, alpha,
                                            _p(shortcut_grad), int(shortcut_accumulate), _p(buf), buf.numel(), _stream()),
           "bn_bwd_apply_sums")


def bn_fold(gamma, beta, mm, mv, eps, scale, shift) -> None:
    _check(load().disyolo_bn_fold(_p(gamma), _p(beta), _p(mm), _p(mv), eps, _p(scale), _p(shift), gamma.numel(),
                                  _stream()), "bn_fold")


def bn_act_fwd(x, scale, shift, residual, y, rows, C_, alpha=0.1) -> None:
    _check(load().disyolo_bn_act_fwd(_p(x), _p(scale), _p(shift), _p(residual), _p(y), rows, C_, alpha, _stream()),
           "bn_act_fwd")


def bn_act_bwd(dy, x, scale, shift, mean, rstd, dx, dgamma, dbeta, rows, C_, ws: Workspace, alpha=0.1,
               shortcut_grad=None, shortcut_accumulate=False) -> None:
    """``shortcut_grad``: the residual shortcut's gradient buffer receives dy (or dy + itself) in the same pass"""
    need = load().disyolo_bn_act_bwd_workspace(rows, C_)
    buf = ws.get(need)
    _check(load().disyolo_bn_act_bwd(_p(dy), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), _p(dx), _p(dgamma),
                                     _p(dbeta), rows, C_, alpha, _p(shortcut_grad), int(shortcut_accumulate), _p(buf),
                                     buf.numel(), _stream()), "bn_act_bwd")


def mask_paste(masks, rects, classids, image_h: int, image_w: int, full_masks, merged) -> None:
    """masks f32 [n,S,S]; rects int32 [n,8]; classids int32 [n]; full_masks uint8 [n,H,W] or None; merged uint8 [H,W]"""
    n = int(rects.shape[0])
    if n:
        _need(masks, torch.float32, "masks")
        _need(rects, torch.int32, "rects")
        _need(classids, torch.int32, "classids")
    _check(load().disyolo_mask_paste(_p(masks) if n else None, n, int(masks.shape[-1]) if n else 1, _p(rects) if n else None,
                                     _p(classids) if n else None, image_h, image_w, _p(full_masks), _p(merged), _stream()),
           "mask_paste")


def letterbox(rgb_u8, out, size: int):
    """rgb_u8 uint8 CUDA [H,W,3] -> out f32 CUDA [size,size,3]; returns the clip window (numpy f32 [4])"""
    import numpy as np
    _need(rgb_u8, torch.uint8, "rgb")
    _need(out, torch.float32, "out")
    if rgb_u8.dim() != 3 or rgb_u8.shape[2] != 3 or tuple(out.shape) != (size, size, 3):
        raise DisyoloError("letterbox: rgb must be [H,W,3] uint8 and out [size,size,3] f32")
    win = (C.c_float * 4)()
    _check(load().disyolo_letterbox(_p(rgb_u8), int(rgb_u8.shape[0]), int(rgb_u8.shape[1]), _p(out), size, win, _stream()),
           "letterbox")
    return np.array(list(win), np.float32)


def polygon_mask(px, py, poly_start, poly_is_out, image_h: int, image_w: int, mask) -> None:
    """px/py f32 CUDA [nv]; poly_start int32 CUDA [npoly+1]; poly_is_out int32 CUDA [npoly]; mask uint8 CUDA [H,W]"""
    _need(px, torch.float32, "px")
    _need(py, torch.float32, "py")
    _need(poly_start, torch.int32, "poly_start")
    _need(poly_is_out, torch.int32, "poly_is_out")
    _need(mask, torch.uint8, "mask")
    _check(load().disyolo_polygon_mask(_p(px), _p(py), _p(poly_start), _p(poly_is_out), int(poly_is_out.numel()), image_h,
                                       image_w, _p(mask), _stream()), "polygon_mask")


def aug_place(src, is_mask: bool, dst, size: int, new_w: int, new_h: int, dx: int, dy: int, flip: int) -> None:
    _need(src, torch.uint8, "src")
    _need(dst, torch.uint8, "dst")
    _check(load().disyolo_aug_place(_p(src), int(is_mask), int(src.shape[0]), int(src.shape[1]), _p(dst), size, new_w, new_h,
                                    dx, dy, flip, _stream()), "aug_place")


PLACE_JOB = np.dtype([("src", np.uint64), ("dst", np.uint64), ("is_mask", np.int32), ("image_h", np.int32), ("image_w", np.int32),
                      ("new_w", np.int32), ("new_h", np.int32), ("dx", np.int32), ("dy", np.int32), ("flip", np.int32)])
assert PLACE_JOB.itemsize == 48      # (disyolo_place_job, include/disyolo.h)


def aug_place_batch(jobs_dev, njobs: int, size: int) -> None:
    """jobs_dev: uint8 CUDA tensor holding njobs PLACE_JOB records (uploaded by the caller on this stream)"""
    _need(jobs_dev, torch.uint8, "jobs")
    if jobs_dev.numel() < njobs * PLACE_JOB.itemsize:
        raise DisyoloError("aug_place_batch: %d jobs do not fit %d bytes" % (njobs, jobs_dev.numel()))
    _check(load().disyolo_aug_place_batch(_p(jobs_dev), njobs, size, _stream()), "aug_place_batch")


def aug_salt_pepper(image, size: int, rows, cols, nsalt: int, npepper: int) -> None:
    _check(load().disyolo_aug_salt_pepper(_p(image), size, _p(rows), _p(cols), nsalt, npepper, _stream()), "aug_salt_pepper")


def aug_change_light(image, size: int, coeff: float) -> None:
    _check(load().disyolo_aug_change_light(_p(image), size, float(coeff), _stream()), "aug_change_light")


def aug_motion_blur3(src, dst, size: int, angle: int, line_type: int) -> None:
    _check(load().disyolo_aug_motion_blur3(_p(src), _p(dst), size, angle, line_type, _stream()), "aug_motion_blur3")


def aug_to_float(image, out) -> None:
    _need(out, torch.float32, "out")
    _check(load().disyolo_aug_to_float(_p(image), _p(out), image.numel(), _stream()), "aug_to_float")


def confusion16(true_map, pred_map, conf) -> None:
    """adds the 4x4 confusion counts of two uint8 CUDA class maps to conf (int64 [16], CUDA)"""
    _need(true_map, torch.uint8, "true_map")
    _need(pred_map, torch.uint8, "pred_map")
    _need(conf, torch.int64, "conf")
    if true_map.numel() != pred_map.numel() or conf.numel() != 16:
        raise DisyoloError("confusion16: shape mismatch")
    _check(load().disyolo_confusion16(_p(true_map), _p(pred_map), true_map.numel(), _p(conf), _stream()), "confusion16")


def upsample2x_bwd(src, dst, B, Hs, Ws, src_C, c_off, C_, accumulate=False) -> None:
    _check(load().disyolo_upsample2x_bwd(_p(src), _p(dst), B, Hs, Ws, src_C, c_off, C_, int(accumulate), _stream()),
           "upsample2x_bwd")


def colsum(x, out, rows, C_, out_C, ws: Workspace) -> None:
    need = load().disyolo_colsum_workspace(rows, C_)
    buf = ws.get(need)
    _check(load().disyolo_colsum(_p(x), _p(out), rows, C_, out_C, _p(buf), buf.numel(), _stream()), "colsum")


def detect(logits3, logits2, logits1, B, S, num_class, anchors_host, clip_window, obj_thresh, nms_thresh, max_det,
           detections, det_count, ws: Workspace) -> None:
    need = load().disyolo_detect_workspace(B, S, num_class)
    buf = ws.get(need)
    anc = (C.c_float * 18)(*[float(v) for v in anchors_host])
    _check(load().disyolo_detect(_p(logits3), _p(logits2), _p(logits1), B, S, num_class, anc, _p(clip_window),
                                 obj_thresh, nms_thresh, max_det, _p(detections), _p(det_count), _p(buf), buf.numel(),
                                 _stream()), "detect")


def yolo_loss(logits, labels, true_boxes, max_boxes, B, S, num_class, anchors_host, ignore_thresh, scales, dlogits,
              losses, ws: Workspace) -> None:
    need = load().disyolo_yolo_loss_workspace(B, S, num_class)
    buf = ws.get(need)
    anc = (C.c_float * 18)(*[float(v) for v in anchors_host])
    lg = (C.c_void_p * 3)(*[_p(t) for t in logits])
    lb = (C.c_void_p * 3)(*[_p(t) for t in labels])
    dl = (C.c_void_p * 3)(*[_p(t) for t in dlogits])
    sc = (C.c_float * 4)(*[float(v) for v in scales])
    _check(load().disyolo_yolo_loss(lg, lb, _p(true_boxes), max_boxes, B, S, num_class, anc, ignore_thresh, sc, dl,
                                    _p(losses), _p(buf), buf.numel(), _stream()), "yolo_loss")


def shuffle_perm(perm_det, perm_gt, B, seed, step_counter) -> None:
    _check(load().disyolo_shuffle_perm(_p(perm_det), perm_det.shape[1], _p(perm_gt), perm_gt.shape[1], B, seed,
                                       _p(step_counter), _stream()), "shuffle_perm")


def mask_rois(detections, max_det, true_boxes, G, perm_det, perm_gt, B, map_size, n_det, n_gt, iou_thresh, rois,
              roi_count) -> None:
    _check(load().disyolo_mask_rois(_p(detections), max_det, _p(true_boxes), G, _p(perm_det), _p(perm_gt), B, map_size,
                                    n_det, n_gt, iou_thresh, _p(rois), _p(roi_count), _stream()), "mask_rois")


def psroi_loss(score, true_masks, G, rois, roi_count, B, map_size, k, mask_scale, dscore, loss, ws: Workspace) -> None:
    need = load().disyolo_psroi_loss_workspace(B, map_size)
    buf = ws.get(need)
    _check(load().disyolo_psroi_loss(_p(score), _p(true_masks), G, _p(rois), _p(roi_count), B, map_size, k, mask_scale,
                                     _p(dscore), _p(loss), _p(buf), buf.numel(), _stream()), "psroi_loss")


def psroi_assemble(score, detections, B, max_det, map_size, k, masks, keep) -> None:
    _check(load().disyolo_psroi_assemble(_p(score), _p(detections), B, max_det, map_size, k, _p(masks), _p(keep),
                                         _stream()), "psroi_assemble")


def adam_step(w, grad, m, v, n, n_decay, lr, b1, b2, eps, l2, t, grad_scale=1.0) -> None:
    _check(load().disyolo_adam_step(_p(w), _p(grad), _p(m), _p(v), n, n_decay, lr, b1, b2, eps, l2, t, grad_scale,
                                    _stream()), "adam_step")


CURRENT_LANE = 0     # lane the recording thread's launches go to (mirrors the executor's state)


def set_lane(lane: int) -> None:
    global CURRENT_LANE
    _check(load().disyolo_cmdlist_set_lane(lane), "cmdlist_set_lane")
    CURRENT_LANE = lane


def lane_sync(src: int, dst: int) -> None:
    _check(load().disyolo_cmdlist_sync(src, dst), "cmdlist_sync")


def lane_mark(lane: int) -> int:
    """remember this point of `lane` in the recording (-1 when not recording)"""
    r = load().disyolo_cmdlist_mark(lane)
    if r < -1:
        _check(r, "cmdlist_mark")
    return r


def lane_wait(mark: Optional[int], lane: int) -> None:
    if mark is not None and mark >= 0:
        _check(load().disyolo_cmdlist_wait(mark, lane), "cmdlist_wait")


def lane_mark_slot(lane: int, slot: int) -> None:
    """named mark that outlives a replay (see lane_wait_slot)"""
    _check(load().disyolo_cmdlist_mark_slot(lane, slot), "cmdlist_mark_slot")


def lane_wait_slot(slot: int, lane: int) -> None:
    """`lane` waits for the point `slot` was last marked at: earlier in this replay or in the previous replay of the list"""
    _check(load().disyolo_cmdlist_wait_slot(slot, lane), "cmdlist_wait_slot")


def adam_step_dev(w, grad, m, v, n, n_decay, lr, b1, b2, eps, l2, step_counter, grad_scale=1.0) -> None:
    _check(load().disyolo_adam_step_dev(_p(w), _p(grad), _p(m), _p(v), n, n_decay, lr, b1, b2, eps, l2,
                                        _p(step_counter), grad_scale, _stream()), "adam_step_dev")


def adam_step_fused(w, grad, m, v, n, n_decay, lr_dev, b1, b2, eps, l2, step_counter, grad_scale, reg_loss_out,
                    ws: Workspace) -> None:
    """Adam with lr and t on the device; also writes 0.5*l2*sum(w_decay^2) of the pre-update weights"""
    _need(lr_dev, torch.float32, "lr_dev")
    need = load().disyolo_adam_fused_workspace(n)
    buf = ws.get(need)
    _check(load().disyolo_adam_step_fused(_p(w), _p(grad), _p(m), _p(v), n, n_decay, _p(lr_dev), b1, b2, eps, l2,
                                          _p(step_counter), grad_scale, _p(reg_loss_out), _p(buf), buf.numel(),
                                          _stream()), "adam_step_fused")


def adam_sweep_parts(n: int) -> int:
    return load().disyolo_adam_sweep_parts(n)


def adam_sweep(w, grad, m, v, n, n_decay, lr_dev, b1, b2, eps, l2, step_counter, grad_scale, parts) -> None:
    """Adam over one slice of the variables (t = *step_counter + 1, counter untouched); l2 partials -> parts"""
    _check(load().disyolo_adam_sweep(_p(w), _p(grad), _p(m), _p(v), n, n_decay, _p(lr_dev), b1, b2, eps, l2,
                                     _p(step_counter), grad_scale, _p(parts), _stream()), "adam_sweep")


def adam_finish(step_counter, parts, nparts, l2, reg_loss_out, record=None) -> None:
    """record = (losses8, mask_loss, reg_loss_in or None, ring): also file the step's total loss into the ring"""
    if record is None:
        _check(load().disyolo_adam_finish(_p(step_counter), _p(parts), nparts, l2, _p(reg_loss_out), _stream()),
               "adam_finish")
        return
    losses8, mask_loss, reg_in, ring = record
    _need(ring, torch.float32, "ring")
    _check(load().disyolo_adam_finish_record(_p(step_counter), _p(parts), nparts, l2, _p(reg_loss_out), _p(losses8),
                                             _p(mask_loss), _p(reg_in), _p(ring), ring.numel(), _stream()), "adam_finish_record")


def add_bf16(src, dst, accumulate: bool) -> None:
    _check(load().disyolo_add_bf16(_p(src), _p(dst), src.numel(), int(accumulate), _stream()), "add_bf16")


def l2_loss(w, n, l2, out, ws: Workspace) -> None:
    need = load().disyolo_l2_workspace(n)
    buf = ws.get(need)
    _check(load().disyolo_l2_loss(_p(w), n, l2, _p(out), _p(buf), buf.numel(), _stream()), "l2_loss")


# ---- gradient exchange as commands of the step (csrc/comm.hip) -------------------------------------------------
COMM_LANE = 3        # the list's lane that carries the collectives and the optimizer sweeps behind them
_DT_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float64: 2}


def reserve_lanes(lanes=(1, COMM_LANE)) -> None:
    """create the side streams of these lanes now (current device).  A data-parallel process calls this BEFORE
    torch.distributed.init_process_group("nccl"): streams created after torch's stream pool exists share hardware
    queues with it, and a lane that shares the caller's stream's queue serialises the step (2.5x, round 5)."""
    mask = 0
    for i in lanes:
        mask |= 1 << int(i)
    _check(load().disyolo_lanes_reserve(mask), "lanes_reserve")


def lanes_report() -> list:
    """how the side lanes that exist on the current device were chosen (csrc/runtime.hip pool_lane): one string per lane"""
    buf = C.create_string_buffer(2048)
    _check(load().disyolo_lanes_report(buf, len(buf)), "lanes_report")
    return [ln for ln in buf.value.decode().splitlines() if ln]


def rccl_path() -> Optional[str]:
    """the RCCL copy this process already maps (torch ships one next to libtorch_hip.so); None = default search"""
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


def comm_load() -> int:
    """dlopen RCCL inside the kernel library; returns RCCL's version code.  Raises DisyoloError when it cannot."""
    p = rccl_path()
    ver = C.c_int(0)
    _check(load().disyolo_comm_load(p.encode() if p else None, C.byref(ver)), "comm_load")
    return ver.value


class Comm:
    """An RCCL communicator owned by the kernel library, one rank per process.  Created collectively: rank 0 makes the
    128-byte id, ``exchange(id_bytes or None) -> id_bytes`` hands it to every rank (torch.distributed's
    broadcast_object_list in dp.py -- any backend, used once), then every rank joins.  The collectives are recordable:
    inside a CmdList recording they become commands of the current lane."""

    def __init__(self, rank: int, world: int, exchange):
        comm_load()
        ident = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            _check(load().disyolo_comm_unique_id(buf), "comm_unique_id")
            ident = buf.raw
        ident = exchange(ident)
        if not isinstance(ident, (bytes, bytearray)) or len(ident) != 128:
            raise DisyoloError("Comm: the id exchange must return rank 0's 128 bytes on every rank")
        h = C.c_void_p()
        _check(load().disyolo_comm_init(C.create_string_buffer(bytes(ident), 128), rank, world, C.byref(h)), "comm_init")
        self.h, self.rank, self.world = h, rank, world

    def close(self) -> None:
        """destroy the communicator (a collective: every rank must call it); device work that uses it is drained first"""
        if getattr(self, "h", None):
            torch.cuda.synchronize()
            _check(load().disyolo_comm_destroy(self.h), "comm_destroy")
            self.h = None

    def __del__(self):
        # a dropped communicator is released like a closed one (its buffers are device memory); at interpreter exit the
        # runtime may already be gone: never raise from here
        try:
            if getattr(self, "h", None) and torch.cuda.is_available():
                self.close()
        except Exception:
            pass

    def allreduce(self, t: torch.Tensor) -> None:
        """t = sum over the ranks (in place), on the current stream / lane"""
        if t.dtype not in _DT_CODE or not t.is_cuda or not t.is_contiguous():
            raise DisyoloError("Comm.allreduce: contiguous CUDA f32 / bf16 / f64 tensor expected")
        _check(load().disyolo_comm_allreduce_sum(self.h, _p(t), t.numel(), _DT_CODE[t.dtype], _stream()), "comm_allreduce_sum")

    def reduce_scatter(self, full: torch.Tensor) -> torch.Tensor:
        """in place: this rank's shard of ``full`` (numel a multiple of the world size) = its sum over the ranks;
        returns the shard view"""
        n = full.numel() // self.world
        if n * self.world != full.numel() or full.dtype not in _DT_CODE or not full.is_contiguous():
            raise DisyoloError("Comm.reduce_scatter: numel must be a multiple of the world size")
        shard = full.view(-1)[self.rank * n:(self.rank + 1) * n]
        _check(load().disyolo_comm_reduce_scatter_sum(self.h, _p(full), _p(shard), n, _DT_CODE[full.dtype], _stream()),
               "comm_reduce_scatter_sum")
        return shard

    def all_gather(self, full: torch.Tensor) -> None:
        """in place: every rank's shard of ``full`` to every rank"""
        n = full.numel() // self.world
        shard = full.view(-1)[self.rank * n:(self.rank + 1) * n]
        _check(load().disyolo_comm_all_gather(self.h, _p(shard), _p(full), n, _DT_CODE[full.dtype], _stream()), "comm_all_gather")


def cast_f32_bf16(src: torch.Tensor, dst: torch.Tensor) -> None:
    _need(src, torch.float32, "src")
    _need(dst, torch.bfloat16, "dst")
    if dst.numel() < src.numel():
        raise DisyoloError("cast_f32_bf16: destination shorter than the source")
    _check(load().disyolo_cast_f32_bf16(_p(src), _p(dst), src.numel(), _stream()), "cast_f32_bf16")


def cast_bf16_f32(src: torch.Tensor, dst: torch.Tensor) -> None:
    _need(src, torch.bfloat16, "src")
    _need(dst, torch.float32, "dst")
    if src.numel() < dst.numel():
        raise DisyoloError("cast_bf16_f32: source shorter than the destination")
    _check(load().disyolo_cast_bf16_f32(_p(src), _p(dst), dst.numel(), _stream()), "cast_bf16_f32")
