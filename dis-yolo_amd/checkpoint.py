"""TensorFlow "tensor bundle" (V2 checkpoint) reader / writer without TensorFlow.

What the reference does with checkpoints:
  * ``tf.train.Saver(var_list=save_list).save(sess, 'output/checkpoint/model.ckpt', global_step=step)``
    every SAVE_ITER steps (train_yolo3_mask.py:58,221-226) -- weights, BN gamma/beta/moving statistics and
    the four biases of ``yolo/convolutional{1..82}``; no optimizer slots, no global_step (:41-58);
  * ``slim.assign_from_checkpoint_fn(WEIGHTS_FILE, include, ignore_missing_vars=True)`` for stage 1
    (:75-107) resp. ``Saver.restore`` of everything for stage 2 (:111) and for testing
    (calculate_test_map.py:184-185).

On disk a checkpoint ``<prefix>`` is
  ``<prefix>.index``                 an SSTable (LevelDB table format, no compression) mapping
                                     "" -> BundleHeaderProto and <variable name> -> BundleEntryProto
  ``<prefix>.data-00000-of-00001``   the tensors' little-endian bytes, back to back in key order
and a text file ``checkpoint`` in the directory names the latest prefix.

The format is restated from TensorFlow's published sources (tensorflow/core/util/tensor_bundle,
tensorflow/core/lib/io/{table_builder,block_builder,format}.cc, tensor_bundle.proto,
lib/hash/crc32c): prefix-compressed keys with restart points, per-block trailer {compression byte,
masked crc32c}, 48-byte footer with the table magic.  **Parity unpinned**: the reference tree ships no
sample checkpoint and TensorFlow cannot be installed here, so the byte layout is checked against a
hand-assembled bundle (tests/test_checkpoint.py), not against TF itself.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
BLOCK_TRAILER = 5                      # compression type (1) + masked crc32c (4)
DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("int32"): DT_INT32, np.dtype("int64"): DT_INT64}


# ---------------------------------------------------------------- crc32c (Castagnoli), masked like leveldb
def _make_table():
    poly = 0x82F63B78
    t = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (poly if c & 1 else 0)
        t[i] = c
    # slicing-by-8 tables
    tabs = [t]
    for k in range(1, 8):
        prev = tabs[-1]
        tabs.append((prev >> 8) ^ t[prev & 0xFF])
    return [x.tolist() for x in tabs]


_T = _make_table()


def _native_crc():
    """the kernel library's host-side crc32c (GB/s) when it has been built; None otherwise"""
    global _NATIVE
    if _NATIVE is None:
        _NATIVE = False
        try:
            from . import lib as L
            if os.path.exists(L.LIB_PATH):
                _NATIVE = L.load().disyolo_crc32c
        except Exception:
            _NATIVE = False
    return _NATIVE or None


_NATIVE = None


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C of ``data``: the library's host routine if present, else the same algorithm in Python
    (file-format arithmetic on the host, not part of the compute path)"""
    fn = _native_crc()
    if fn is not None and len(data) >= 64:
        buf = np.frombuffer(data, dtype=np.uint8)
        return int(fn(buf.ctypes.data, len(data), crc))
    return crc32c_py(data, crc)


def crc32c_py(data: bytes, crc: int = 0) -> int:
    c = crc ^ 0xFFFFFFFF
    mv = memoryview(data)
    n = len(mv)
    i = 0
    t0, t1, t2, t3, t4, t5, t6, t7 = _T
    if n >= 8:
        words = np.frombuffer(mv[: n - n % 8], dtype="<u4").tolist()
        for j in range(0, len(words), 2):
            a = words[j] ^ c
            b = words[j + 1]
            c = (t7[a & 0xFF] ^ t6[(a >> 8) & 0xFF] ^ t5[(a >> 16) & 0xFF] ^ t4[a >> 24] ^
                 t3[b & 0xFF] ^ t2[(b >> 8) & 0xFF] ^ t1[(b >> 16) & 0xFF] ^ t0[b >> 24])
        i = n - n % 8
    for k in range(i, n):
        c = t0[(c ^ mv[k]) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def unmask_crc(m: int) -> int:
    r = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------- varints / minimal protobuf
def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _read_varint(b: bytes, pos: int) -> Tuple[int, int]:
    shift, val = 0, 0
    while True:
        c = b[pos]
        pos += 1
        val |= (c & 0x7F) << shift
        if c < 0x80:
            return val, pos
        shift += 7


def _field(num: int, wire: int, payload: bytes) -> bytes:
    return _varint((num << 3) | wire) + payload


def _parse(b: bytes) -> List[Tuple[int, int, object]]:
    """[(field number, wire type, value)]: varint -> int, fixed32 -> int, length-delimited -> bytes"""
    out, pos = [], 0
    while pos < len(b):
        tag, pos = _read_varint(b, pos)
        num, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _read_varint(b, pos)
        elif wire == 2:
            ln, pos = _read_varint(b, pos)
            v = b[pos:pos + ln]
            pos += ln
        elif wire == 5:
            v = struct.unpack_from("<I", b, pos)[0]
            pos += 4
        elif wire == 1:
            v = struct.unpack_from("<Q", b, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        out.append((num, wire, v))
    return out


def encode_header(num_shards: int = 1) -> bytes:
    """BundleHeaderProto {num_shards = 1; endianness = LITTLE (default, omitted); version {producer: 1}}"""
    version = _field(1, 0, _varint(1))
    return _field(1, 0, _varint(num_shards)) + _field(3, 2, _varint(len(version)) + version)


def encode_entry(dtype: int, shape: Iterable[int], offset: int, size: int, crc_masked: int, shard_id: int = 0) -> bytes:
    """BundleEntryProto; zero-valued scalar fields are omitted like proto3 serialisation does"""
    dims = b"".join(_field(2, 2, _varint(len(d)) + d) for d in (_field(1, 0, _varint(int(s))) for s in shape))
    out = _field(1, 0, _varint(dtype)) + _field(2, 2, _varint(len(dims)) + dims)
    if shard_id:
        out += _field(3, 0, _varint(shard_id))
    if offset:
        out += _field(4, 0, _varint(offset))
    out += _field(5, 0, _varint(size))
    out += _field(6, 5, struct.pack("<I", crc_masked))
    return out


def decode_entry(b: bytes) -> Dict:
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": 0}
    for num, wire, v in _parse(b):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            for n2, _, d in _parse(v):
                if n2 == 2:
                    size = 0
                    for n3, _, x in _parse(d):
                        if n3 == 1:
                            size = x
                    e["shape"].append(size)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["slices"] += 1
    return e


# ---------------------------------------------------------------- SSTable blocks
def build_block(items: List[Tuple[bytes, bytes]], restart_interval: int) -> bytes:
    """entries {shared, non_shared, value_len, key suffix, value}; restart offsets; restart count"""
    out = bytearray()
    restarts = []
    last = b""
    for i, (k, v) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            for a, b_ in zip(last, k):
                if a != b_:
                    break
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def parse_block(b: bytes) -> List[Tuple[bytes, bytes]]:
    nrestart = struct.unpack_from("<I", b, len(b) - 4)[0]
    end = len(b) - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _read_varint(b, pos)
        non_shared, pos = _read_varint(b, pos)
        vlen, pos = _read_varint(b, pos)
        key = key[:shared] + b[pos:pos + non_shared]
        pos += non_shared
        out.append((key, b[pos:pos + vlen]))
        pos += vlen
    return out


def _with_trailer(block: bytes) -> bytes:
    t = b"\x00"                                          # kNoCompression
    return block + t + struct.pack("<I", mask_crc(crc32c(block + t)))


def build_table(items: List[Tuple[bytes, bytes]], block_size: int = 4096, restart_interval: int = 16) -> bytes:
    """keys must be sorted; data blocks of ~block_size, empty metaindex block, index block (restart
    interval 1, key = last key of the block, value = block handle), footer"""
    out = bytearray()
    index: List[Tuple[bytes, bytes]] = []
    cur: List[Tuple[bytes, bytes]] = []
    cur_bytes = 0

    def flush():
        nonlocal cur, cur_bytes
        if not cur:
            return
        blk = build_block(cur, restart_interval)
        index.append((cur[-1][0], _varint(len(out)) + _varint(len(blk))))
        out.extend(_with_trailer(blk))
        cur, cur_bytes = [], 0

    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 3
        if cur_bytes >= block_size:
            flush()
    flush()
    meta = build_block([], 1)
    meta_handle = _varint(len(out)) + _varint(len(meta))
    out.extend(_with_trailer(meta))
    idx = build_block(index, 1)
    idx_handle = _varint(len(out)) + _varint(len(idx))
    out.extend(_with_trailer(idx))
    footer = meta_handle + idx_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    return bytes(out)


def parse_table(b: bytes, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    if len(b) < 48 or struct.unpack_from("<Q", b, len(b) - 8)[0] != TABLE_MAGIC:
        raise ValueError("not a TensorFlow checkpoint index (bad table magic)")
    f = b[len(b) - 48:]
    _, p = _read_varint(f, 0)
    _, p = _read_varint(f, p)
    ioff, p = _read_varint(f, p)
    isz, p = _read_varint(f, p)

    def block(off, size):
        raw = b[off:off + size + BLOCK_TRAILER]
        if raw[size] != 0:
            raise ValueError("compressed checkpoint index blocks are not supported")
        if verify and unmask_crc(struct.unpack_from("<I", raw, size + 1)[0]) != crc32c(raw[:size + 1]):
            raise ValueError("checkpoint index block fails its crc32c")
        return raw[:size]

    items = []
    for _, handle in parse_block(block(ioff, isz)):
        off, q = _read_varint(handle, 0)
        size, _ = _read_varint(handle, q)
        items.extend(parse_block(block(off, size)))
    return items


# ---------------------------------------------------------------- the bundle
def save_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], update_state_file: bool = True) -> None:
    """``Saver.save``: writes <prefix>.index and <prefix>.data-00000-of-00001 (float32 / int32 / int64
    variables) and, like TF, the ``checkpoint`` state file naming the latest prefix."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    names = sorted(tensors, key=lambda s: s.encode())
    items = [(b"", encode_header(1))]
    offset = 0
    with open(prefix + ".data-00000-of-00001.tmp", "wb") as f:
        for name in names:
            a = np.asarray(tensors[name])
            if not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)           # (np.ascontiguousarray would turn a 0-d variable into shape [1])
            if a.dtype not in _DT_OF:
                raise TypeError("variable %s has unsupported dtype %s" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            items.append((name.encode(), encode_entry(_DT_OF[a.dtype], a.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    with open(prefix + ".index.tmp", "wb") as f:
        f.write(build_table(items))
    os.replace(prefix + ".data-00000-of-00001.tmp", prefix + ".data-00000-of-00001")
    os.replace(prefix + ".index.tmp", prefix + ".index")
    if update_state_file:
        base = os.path.basename(prefix)
        state = os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint")
        olds = []
        if os.path.exists(state):
            for ln in open(state):
                if ln.startswith("all_model_checkpoint_paths:"):
                    olds.append(ln.split('"')[1])
        olds = [o for o in olds if o != base] + [base]
        with open(state, "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % base)
            for o in olds:
                f.write('all_model_checkpoint_paths: "%s"\n' % o)


def list_variables(prefix: str) -> Dict[str, Tuple[Tuple[int, ...], int]]:
    """name -> (shape, dtype enum), like tf.train.list_variables"""
    items = parse_table(open(prefix + ".index", "rb").read())
    out = {}
    for k, v in items:
        if k == b"":
            continue
        e = decode_entry(v)
        out[k.decode()] = (tuple(e["shape"]), e["dtype"])
    return out


def load_checkpoint(prefix: str, names: Optional[Iterable[str]] = None, verify: bool = True) -> Dict[str, np.ndarray]:
    """``Saver.restore`` / ``assign_from_checkpoint_fn``: name -> array.  ``names`` restricts the read
    (missing ones are skipped: ignore_missing_vars=True, train_yolo3_mask.py:104-105)."""
    items = dict(parse_table(open(prefix + ".index", "rb").read(), verify))
    hdr = {n: v for n, _, v in _parse(items.get(b"", b""))}
    nshards = hdr.get(1, 1)
    if hdr.get(2, 0) != 0:
        raise ValueError("big-endian checkpoints are not supported")
    want = None if names is None else {n.encode() for n in names}
    files = {}
    out = {}
    for k, v in items.items():
        if k == b"" or (want is not None and k not in want):
            continue
        e = decode_entry(v)
        if e["slices"]:
            raise ValueError("partitioned variable %s: sliced checkpoint entries are not supported" % k.decode())
        if e["dtype"] not in _DTYPES:
            raise TypeError("variable %s has unsupported dtype enum %d" % (k.decode(), e["dtype"]))
        sid = e["shard_id"]
        if sid not in files:
            files[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, nshards), "rb")
        f = files[sid]
        f.seek(e["offset"])
        raw = f.read(e["size"])
        if len(raw) != e["size"]:
            raise ValueError("checkpoint data file is truncated at variable %s" % k.decode())
        if verify and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("variable %s fails its crc32c" % k.decode())
        out[k.decode()] = np.frombuffer(raw, dtype=_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    for f in files.values():
        f.close()
    return out


def latest_checkpoint(directory: str) -> Optional[str]:
    """tf.train.latest_checkpoint"""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    for ln in open(state):
        if ln.startswith("model_checkpoint_path:"):
            p = ln.split('"')[1]
            return p if os.path.isabs(p) else os.path.join(directory, p)
    return None


# ---------------------------------------------------------------- the reference's variable sets
def stage1_include_names() -> List[str]:
    """the ``include`` list of Solver.__init__ (train_yolo3_mask.py:75-103): conv+BN variables of layers
    1-58, 60-66, 68-74 and weights+biases of the detection convs 59/67/75 -- NOT the mask subnet 76-82"""
    out = []
    for i in list(range(1, 59)) + list(range(60, 67)) + list(range(68, 75)):
        base = "yolo/convolutional%d/" % i
        out += [base + "weights", base + "BatchNorm/beta", base + "BatchNorm/gamma", base + "BatchNorm/moving_mean",
                base + "BatchNorm/moving_variance"]
    for i in (59, 67, 75):
        out += ["yolo/convolutional%d/weights" % i, "yolo/convolutional%d/biases" % i]
    return out


def save_net(net, prefix: str) -> None:
    """``self.saver.save`` (train_yolo3_mask.py:221-226): every variable of ``yolo/convolutional{1..82}``"""
    save_checkpoint(prefix, {k: v.detach().cpu().numpy() for k, v in net.params.items()})


def restore_net(net, prefix: str, stage1_include: bool = False) -> List[str]:
    """stage1_include=True: ``assign_from_checkpoint_fn(include, ignore_missing_vars=True)`` -- variables in
    the include list that the file holds (shape mismatches raise, as TF does); False: ``Saver.restore`` of
    every variable (a missing one raises).  Returns the restored names."""
    names = stage1_include_names() if stage1_include else list(net.params)
    got = load_checkpoint(prefix, names)
    if not stage1_include:
        missing = [n for n in names if n not in got]
        if missing:
            raise KeyError("checkpoint %s lacks %d variables, e.g. %s" % (prefix, len(missing), missing[:3]))
    for n, a in got.items():
        if n in net.params and tuple(a.shape) != tuple(net.params[n].shape):
            raise ValueError("variable %s: checkpoint shape %s, graph shape %s" % (n, a.shape, tuple(net.params[n].shape)))
    net.load_state_dict({n: a for n, a in got.items() if n in net.params}, strict=False)
    return sorted(got)
