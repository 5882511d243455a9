"""TF tensor-bundle checkpoint I/O (dis-yolo_amd/checkpoint.py) -- CPU only.

Parity with TensorFlow itself is UNPINNED: the reference ships no checkpoint and TF is not installable
here.  What is pinned: CRC-32C by the RFC 3720 vectors; the byte layout by a bundle assembled BY HAND in
this file from the format description (explicit protobuf / block / footer bytes), which the writer must
reproduce byte for byte and the reader must parse; the variable set / names / shapes by the reference's
layout (train_yolo3_mask.py:86-103; SURVEY.md section 5: 398 variables, 61,709,041 floats)."""
import os
import struct

import numpy as np
import pytest
import torch

from disyolo_amd import checkpoint as ck
from disyolo_amd.net import YOLONet


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4
    assert ck.crc32c_py(b"123456789") == 0xE3069283
    assert ck.crc32c_py(bytes(32)) == 0x8A9136AA
    assert ck.crc32c_py(bytes([0xFF] * 32)) == 0x62A8AB43
    assert ck.crc32c_py(bytes(range(32))) == 0x46DD794E
    data = os.urandom(4099)
    assert ck.crc32c(data) == ck.crc32c_py(data)                         # native host routine == Python
    assert ck.crc32c_py(data[100:], ck.crc32c_py(data[:100])) == ck.crc32c_py(data)   # incremental
    # leveldb's mask: rotate right 15, add the constant
    assert ck.mask_crc(0) == 0xA282EAD8 and ck.unmask_crc(ck.mask_crc(0xDEADBEEF)) == 0xDEADBEEF


def _trailer(block: bytes) -> bytes:
    return block + b"\x00" + struct.pack("<I", ck.mask_crc(ck.crc32c_py(block + b"\x00")))


def test_writer_reproduces_a_hand_assembled_two_tensor_bundle(tmp_path):
    a = np.array([1.0, 2.0], np.float32)
    bc = np.array([[7, -1]], np.int32)
    data = a.tobytes() + bc.tobytes()                       # tensors back to back in key order: "a" < "b/c"
    crc_a = struct.pack("<I", ck.mask_crc(ck.crc32c_py(a.tobytes())))
    crc_bc = struct.pack("<I", ck.mask_crc(ck.crc32c_py(bc.tobytes())))
    # BundleHeaderProto: num_shards(1)=1, version(3){producer(1)=1}
    header = bytes.fromhex("0801" "1a020801")
    # BundleEntryProto "a": dtype(1)=DT_FLOAT(1); shape(2){dim(2){size(1)=2}}; size(5)=8; crc32c(6) fixed32
    e_a = bytes.fromhex("0801" "1204" "1202" "0802" "2808" "35") + crc_a
    # "b/c": dtype=DT_INT32(3); shape {dim{1}, dim{2}}; offset(4)=8; size=8; crc
    e_bc = bytes.fromhex("0803" "1208" "12020801" "12020802" "2008" "2808" "35") + crc_bc
    # data block: {shared, non_shared, value_len, key suffix, value}*, restart array [0], restart count 1
    blk = (bytes([0, 0, len(header)]) + header +
           bytes([0, 1, len(e_a)]) + b"a" + e_a +
           bytes([0, 3, len(e_bc)]) + b"b/c" + e_bc +
           struct.pack("<II", 0, 1))
    table = _trailer(blk)
    meta = struct.pack("<II", 0, 1)                          # empty metaindex block
    meta_off = len(table)
    table += _trailer(meta)
    # index block: one entry, key = last key of the data block, value = handle {offset 0, size len(blk)}
    handle = bytes([0, len(blk)])
    assert len(blk) < 128
    idx = bytes([0, 3, len(handle)]) + b"b/c" + handle + struct.pack("<II", 0, 1)
    idx_off = len(table)
    table += _trailer(idx)
    footer = bytes([meta_off, len(meta)]) + ck._varint(idx_off) + bytes([len(idx)])
    footer += b"\x00" * (40 - len(footer)) + bytes.fromhex("57fb808b247547db")
    table += footer
    assert len(footer) == 48

    prefix = str(tmp_path / "model.ckpt-500")
    ck.save_checkpoint(prefix, {"b/c": bc, "a": a})
    assert open(prefix + ".data-00000-of-00001", "rb").read() == data
    assert open(prefix + ".index", "rb").read() == table
    # ... and the reader takes the hand-made files
    hand = str(tmp_path / "hand.ckpt")
    open(hand + ".index", "wb").write(table)
    open(hand + ".data-00000-of-00001", "wb").write(data)
    got = ck.load_checkpoint(hand)
    assert set(got) == {"a", "b/c"} and np.array_equal(got["a"], a) and np.array_equal(got["b/c"], bc)
    assert ck.list_variables(hand) == {"a": ((2,), ck.DT_FLOAT), "b/c": ((1, 2), ck.DT_INT32)}
    assert ck.latest_checkpoint(str(tmp_path)) == prefix
    assert 'model_checkpoint_path: "model.ckpt-500"' in open(tmp_path / "checkpoint").read()


def test_prefix_compression_restarts_and_multiple_blocks(tmp_path):
    """398 variable names share long prefixes: keys are delta-encoded with a restart every 16 entries and
    the table spills into several data blocks; every name must come back"""
    rng = np.random.RandomState(0)
    tensors = {"yolo/convolutional%d/BatchNorm/%s" % (i, leaf): rng.randn(7).astype(np.float32)
               for i in range(1, 83) for leaf in ("beta", "gamma", "moving_mean", "moving_variance")}
    prefix = str(tmp_path / "m")
    ck.save_checkpoint(prefix, tensors)
    raw = open(prefix + ".index", "rb").read()
    items = ck.parse_table(raw)
    assert [k for k, _ in items] == sorted([b""] + [n.encode() for n in tensors])
    assert len(raw) < sum(len(n) + 40 for n in tensors) * 0.8            # the shared prefixes were not stored
    # a scalar (global_step-like) and an empty variable keep their shapes: [] and [0], not [1]
    tensors["global_step"] = np.array(7, np.int64)
    tensors["yolo/empty"] = np.zeros((0,), np.float32)
    tensors["yolo/strided"] = np.arange(12, dtype=np.float32).reshape(3, 4).T       # non-contiguous input
    ck.save_checkpoint(prefix, tensors)
    got = ck.load_checkpoint(prefix)
    assert all(np.array_equal(got[n], tensors[n]) for n in tensors)
    assert got["global_step"].shape == () and int(got["global_step"]) == 7 and got["yolo/empty"].shape == (0,)
    assert ck.list_variables(prefix)["global_step"] == ((), ck.DT_INT64) and got["yolo/strided"].shape == (4, 3)
    # corruption is detected: one flipped data byte, one flipped index byte
    d = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    d[100] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(d))
    with pytest.raises(ValueError, match="crc32c"):
        ck.load_checkpoint(prefix)
    r = bytearray(raw)
    r[50] ^= 1
    open(prefix + ".index", "wb").write(bytes(r))
    with pytest.raises(ValueError, match="crc32c"):
        ck.load_checkpoint(prefix)


def test_network_round_trip_and_stage1_include_list(tmp_path):
    """Saver.save of every yolo/convolutional{1..82} variable and the two restore modes of Solver.__init__"""
    src = YOLONet(training=True, stage=2, seed=3, plan_only=True)
    with torch.no_grad():
        for i, (n, t) in enumerate(sorted(src.params.items())):
            if "BatchNorm" in n or n.endswith("biases"):
                t.add_(0.01 * (i % 7) + 0.5)                  # make every variable distinguishable from its init
    prefix = str(tmp_path / "model.ckpt-1000")
    ck.save_net(src, prefix)
    listed = ck.list_variables(prefix)
    assert len(listed) == 82 + 78 * 4 + 4 == 398
    assert sum(int(np.prod(s)) for s, _ in listed.values()) == 61_709_041
    assert listed["yolo/convolutional1/weights"] == ((3, 3, 3, 32), ck.DT_FLOAT)        # HWIO
    assert listed["yolo/convolutional61/weights"] == ((1, 1, 768, 256), ck.DT_FLOAT)    # [skip5 | upsampled]
    assert listed["yolo/convolutional82/biases"] == ((9,), ck.DT_FLOAT)
    assert "yolo/convolutional59/BatchNorm/gamma" not in listed and os.path.getsize(prefix + ".data-00000-of-00001") == 61_709_041 * 4
    # Saver.restore: everything, bit for bit
    dst = YOLONet(training=True, stage=2, seed=9, plan_only=True)
    names = ck.restore_net(dst, prefix)
    assert len(names) == 398 and all(torch.equal(dst.params[n], src.params[n]) for n in src.params)
    # stage 1: the include list leaves the mask subnet (76-82) at its fresh initialisation
    fresh = YOLONet(training=True, stage=1, seed=9, plan_only=True)
    before = {n: t.clone() for n, t in fresh.params.items()}
    names = ck.restore_net(fresh, prefix, stage1_include=True)
    assert len(names) == 72 * 5 + 3 * 2
    for n in fresh.params:
        layer = int(n.split("convolutional")[1].split("/")[0])
        if layer >= 76 or layer in (59, 67, 75) and "BatchNorm" in n:
            assert torch.equal(fresh.params[n], before[n]), n
        else:
            assert torch.equal(fresh.params[n], src.params[n]), n
    # a checkpoint that lacks a variable: Saver.restore raises, the include-list restore ignores it
    part = {k: v.detach().numpy() for k, v in src.params.items() if "convolutional10/" not in k}
    ck.save_checkpoint(str(tmp_path / "partial"), part)
    with pytest.raises(KeyError):
        ck.restore_net(dst, str(tmp_path / "partial"))
    assert len(ck.restore_net(fresh, str(tmp_path / "partial"), stage1_include=True)) == 71 * 5 + 6


def test_reader_survives_truncated_and_bit_flipped_bundles(tmp_path):
    """Fuzz loop over a written bundle (sanitizer row -- the SSTable reader is Python, so the property is
    "fails cleanly"): every truncation of the .index and 300 single-bit flips of .index / .data must either load the exact
    tensors (a flip in padding) or raise a Python exception -- never hang, never return silently different data while
    verify=True (block trailers and tensors carry masked crc32c)."""
    import random
    rng = np.random.RandomState(5)
    tensors = {"yolo/convolutional%d/weights" % i: rng.randn(3, 3, 4, 5 + i).astype(np.float32) for i in range(1, 40)}
    tensors["yolo/convolutional1/BatchNorm/gamma"] = rng.randn(7).astype(np.float32)
    prefix = str(tmp_path / "model.ckpt-7")
    ck.save_checkpoint(prefix, tensors)
    index = open(prefix + ".index", "rb").read()
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    want = ck.load_checkpoint(prefix)
    assert set(want) == set(tensors)

    def attempt():
        try:
            got = ck.load_checkpoint(prefix)
        except Exception:
            return "raised"
        assert set(got) == set(want) and all(np.array_equal(got[k], want[k]) for k in want), "silently different data"
        return "same"

    outcomes = {"raised": 0, "same": 0}
    for cut in list(range(0, len(index), max(1, len(index) // 120))) + [len(index) - 1]:
        open(prefix + ".index", "wb").write(index[:cut])
        outcomes[attempt()] += 1
    r = random.Random(11)
    for _ in range(200):
        b = bytearray(index)
        pos = r.randrange(len(b))
        b[pos] ^= 1 << r.randrange(8)
        open(prefix + ".index", "wb").write(bytes(b))
        outcomes[attempt()] += 1
    open(prefix + ".index", "wb").write(index)
    for _ in range(100):
        b = bytearray(data)
        pos = r.randrange(len(b))
        b[pos] ^= 1 << r.randrange(8)
        open(prefix + ".data-00000-of-00001", "wb").write(bytes(b))
        outcomes[attempt()] += 1
    open(prefix + ".data-00000-of-00001", "wb").write(data[:len(data) // 2])
    outcomes[attempt()] += 1
    assert outcomes["raised"] >= 300, outcomes          # (a handful of flips land in bytes no reader looks at)
