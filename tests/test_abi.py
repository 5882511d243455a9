"""The C-ABI shared library loads without a GPU and exports every symbol that
include/disyolo.h declares; the ctypes table in lib.py covers the same set."""
import ctypes
import os
import re

import pytest

from disyolo_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "disyolo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(disyolo_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("disyolo_conv2d_fwd", "disyolo_conv2d_wgrad", "disyolo_bn_finalize", "disyolo_bn_act_bwd",
                 "disyolo_detect", "disyolo_yolo_loss", "disyolo_psroi_loss", "disyolo_psroi_assemble",
                 "disyolo_adam_step", "disyolo_cmdlist_run", "disyolo_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        pytest.fail("libdisyolo_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), "%s declared in include/disyolo.h but not exported" % name


def test_ctypes_table_matches_header():
    assert sorted(L.EXPORTS) == declared_functions()
    assert L.load().disyolo_version() >= 100
    # 16 x int32/float + 8 pointers, + 6 pointers and a float of the batch-norm backward epilogue, + 2 floats (padded) and
    # 12 pointers of the in-launch batch norm (round 6)
    assert ctypes.sizeof(L.ConvDesc) == L.load().disyolo_conv_desc_size() == 288


def test_argument_errors_are_reported_not_thrown():
    # no GPU needed: validation happens before any launch
    lib = L.load()
    d = L.ConvDesc()
    assert lib.disyolo_conv2d_fwd(ctypes.byref(d), None) == -1
    assert b"conv" in lib.disyolo_last_error()
    assert lib.disyolo_cmdlist_run(None, 0, 0, None) == -1
    assert lib.disyolo_adam_step(None, None, None, None, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 1.0, None) == -1
    assert lib.disyolo_detect_workspace(8, 576, 3) > 8 * 20412 * 24


def test_exchange_entry_points_validate_and_resolve_rccl_without_a_gpu():
    """csrc/comm.hip: RCCL is resolved at run time (no link-time dependency: the library above loaded without it being named),
    argument errors come back as codes, and nothing here computes or needs a device"""
    lib = L.load()
    assert L.comm_load() >= 20000                     # RCCL's version code (2.x.y -> 2xxyy), from the copy torch ships
    buf = ctypes.create_string_buffer(64)
    assert lib.disyolo_comm_allreduce_sum(None, buf, 16, 0, None) == -1 and b"comm_allreduce_sum" in lib.disyolo_last_error()
    assert lib.disyolo_comm_allreduce_sum(buf, buf, 16, 7, None) == -1              # unknown dtype code
    assert lib.disyolo_comm_init(buf, 3, 2, None) == -1                               # rank >= nranks / null output
    assert lib.disyolo_comm_destroy(None) == 0
    assert lib.disyolo_cast_f32_bf16(None, buf, 4, None) == -1
    assert lib.disyolo_cmdlist_count(None, 0, 0) == -1
    assert lib.disyolo_cmdlist_lane_stream(None, 1) is None
