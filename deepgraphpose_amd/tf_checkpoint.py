"""TensorFlow-1.x checkpoint reader / writer without TensorFlow.

The reference restores and saves its models with `tf.train.Saver` (DGP/models/fitdgp.py:132-152,394-401,689-696,
830-839; DGP/models/eval.py:194-211).  Two on-disk formats occur:

  V2 "tensor bundle"   <prefix>.index  (an SSTable: key "" -> BundleHeaderProto, key <var name> -> BundleEntryProto
                       {dtype, shape, shard_id, offset, size, crc32c}) + <prefix>.data-00000-of-00001 (raw little-endian
                       tensor bytes).  This is what Saver writes for snapshot-step{k}-final--0.
  V1 "tensor slice"    one file (e.g. the ImageNet resnet_v1_50.ckpt): an SSTable whose values are SavedTensorSlices
                       protos; key "" holds the meta list, the other entries hold one slice of data each.

Both sit on TensorFlow's copy of the LevelDB table format (tensorflow/core/lib/io/{table,block,format}.cc): data
blocks of prefix-compressed (shared, non_shared, value_len, key_delta, value) entries with a restart array, each
block followed by a 1-byte compression type (0 = none, 1 = snappy) and a masked crc32c; a metaindex block, an index
block of BlockHandles, and a 48-byte footer ending in the magic 0xdb4775248b80fb57.

Written from the published format.  TensorFlow is not installable in the build container, so conformance is
checked by round trips through the writer in this file (tests/test_tf_checkpoint_cpu.py), NOT against files
produced by TensorFlow itself -- treat the first real checkpoint you load as the acceptance test.
Only fp32 (DT_FLOAT) tensors are handled; that is all the hot path stores.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterator, List, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT = 1


# ------------------------------------------------------------------------------------------- crc32c (Castagnoli)
def _make_crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC = _make_crc_table()


def crc32c(data: bytes, crc: int = 0) -> int:
    if len(data) >= 4096:                             # big tensors: the C helper in libdgp_hip.so when it is built
        try:
            from . import _lib
            buf = np.frombuffer(data, dtype=np.uint8)
            return int(_lib.load().dgp_crc32c(buf.ctypes.data, buf.size, crc))
        except Exception:
            pass
    crc ^= 0xFFFFFFFF
    for b in data:
        crc = _CRC[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------- varints / mini-protobuf
def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """Yield (field number, wire type, value) of one protobuf message (value: int or bytes)."""
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _pb_key(fn: int, wt: int) -> bytes:
    return _put_varint((fn << 3) | wt)


def _pb_bytes(fn: int, b: bytes) -> bytes:
    return _pb_key(fn, 2) + _put_varint(len(b)) + b


def _pb_int(fn: int, v: int) -> bytes:
    return _pb_key(fn, 0) + _put_varint(v)


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for fn, _, v in _pb_fields(buf):
        if fn == 2:                                   # TensorShapeProto.dim
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = v2
            dims.append(size)
    return tuple(dims)


def _encode_shape(shape) -> bytes:
    return b"".join(_pb_bytes(2, _pb_int(1, int(d))) for d in shape)


# ------------------------------------------------------------------------------------------- snappy (decode only)
def _snappy_decompress(data: bytes) -> bytes:
    n, pos = _get_varint(data, 0)
    out = bytearray()
    while pos < len(data):
        tag = data[pos]
        pos += 1
        t = tag & 3
        if t == 0:                                    # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += data[pos:pos + ln]
            pos += ln
            continue
        if t == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | data[pos]
            pos += 1
        elif t == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 4], "little")
            pos += 4
        for _ in range(ln):                           # may overlap
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


# ------------------------------------------------------------------------------------------- SSTable
def _read_block(f, offset: int, size: int, verify: bool = True) -> bytes:
    f.seek(offset)
    raw = f.read(size + 5)
    body, ctype, crc = raw[:size], raw[size], struct.unpack("<I", raw[size + 1:size + 5])[0]
    if verify and mask_crc(crc32c(raw[:size + 1])) != crc:
        raise ValueError("SSTable block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise ValueError("unknown block compression %d" % ctype)
    return body


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    num_restarts = struct.unpack("<I", block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def table_entries(path: str, verify: bool = True) -> Iterator[Tuple[bytes, bytes]]:
    """All (key, value) pairs of an SSTable file in key order."""
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if size < 48:
            raise ValueError("%s is too short to be an SSTable" % path)
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack("<Q", footer[40:])[0] != TABLE_MAGIC:
            raise ValueError("%s: bad table magic (not a TensorFlow checkpoint table)" % path)
        pos = 0
        _, pos = _get_varint(footer, pos)             # metaindex handle
        _, pos = _get_varint(footer, pos)
        ioff, pos = _get_varint(footer, pos)
        isz, pos = _get_varint(footer, pos)
        for _, handle in _block_entries(_read_block(f, ioff, isz, verify)):
            boff, p2 = _get_varint(handle, 0)
            bsz, _ = _get_varint(handle, p2)
            yield from _block_entries(_read_block(f, boff, bsz, verify))


class _TableWriter:
    """Minimal uncompressed SSTable writer (one entry per restart interval of 16, 4 KiB blocks)."""

    def __init__(self, path: str):
        self.f = open(path, "wb")
        self.off = 0
        self.index: List[Tuple[bytes, int, int]] = []
        self.cur: List[Tuple[bytes, bytes]] = []
        self.cur_size = 0

    def _emit_block(self, entries) -> Tuple[int, int]:
        body, restarts, last = bytearray(), [], b""
        for i, (k, v) in enumerate(entries):
            shared = 0
            if i % 16 == 0:
                restarts.append(len(body))
            else:
                while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                    shared += 1
            body += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
            last = k
        if not restarts:
            restarts = [0]
        body += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
        trailer = b"\x00" + struct.pack("<I", mask_crc(crc32c(bytes(body) + b"\x00")))
        start = self.off
        self.f.write(body + trailer)
        self.off += len(body) + 5
        return start, len(body)

    def add(self, key: bytes, value: bytes):
        self.cur.append((key, value))
        self.cur_size += len(key) + len(value)
        if self.cur_size >= 4096:
            self._flush()

    def _flush(self):
        if self.cur:
            off, sz = self._emit_block(self.cur)
            self.index.append((self.cur[-1][0], off, sz))
            self.cur, self.cur_size = [], 0

    def close(self):
        self._flush()
        moff, msz = self._emit_block([])
        ioff, isz = self._emit_block([(k, _put_varint(o) + _put_varint(s)) for k, o, s in self.index])
        footer = _put_varint(moff) + _put_varint(msz) + _put_varint(ioff) + _put_varint(isz)
        footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
        self.f.write(footer)
        self.f.close()


# ------------------------------------------------------------------------------------------- V2 bundles
def read_v2(prefix: str, verify: bool = False) -> Dict[str, np.ndarray]:
    """<prefix>.index + <prefix>.data-0000S-of-0000N -> {variable name: fp32 array}."""
    out, shards, num_shards = {}, {}, 1
    entries = []
    for key, val in table_entries(prefix + ".index"):
        if key == b"":
            for fn, _, v in _pb_fields(val):          # BundleHeaderProto
                if fn == 1:
                    num_shards = v
                elif fn == 2 and v != 0:
                    raise ValueError("big-endian bundles are not supported")
            continue
        e = dict(dtype=0, shape=(), shard=0, offset=0, size=0, crc=None, sliced=False)
        for fn, wt, v in _pb_fields(val):             # BundleEntryProto
            if fn == 1:
                e["dtype"] = v
            elif fn == 2:
                e["shape"] = _parse_shape(v)
            elif fn == 3:
                e["shard"] = v
            elif fn == 4:
                e["offset"] = v
            elif fn == 5:
                e["size"] = v
            elif fn == 6:
                e["crc"] = struct.unpack("<I", v)[0]
            elif fn == 7:
                e["sliced"] = True
        entries.append((key.decode(), e))
    for name, e in entries:
        if e["sliced"]:
            raise ValueError("partitioned variable %s is not supported" % name)
        if e["dtype"] != DT_FLOAT:
            continue                                  # e.g. int64 global_step
        if e["shard"] not in shards:
            shards[e["shard"]] = open("%s.data-%05d-of-%05d" % (prefix, e["shard"], num_shards), "rb")
        f = shards[e["shard"]]
        f.seek(e["offset"])
        raw = f.read(e["size"])
        if verify and e["crc"] is not None and mask_crc(crc32c(raw)) != e["crc"]:
            raise ValueError("tensor checksum mismatch for %s" % name)
        out[name] = np.frombuffer(raw, dtype="<f4").reshape(e["shape"]).copy()
    for f in shards.values():
        f.close()
    return out


def write_v2(prefix: str, tensors: Dict[str, np.ndarray]):
    """Write a single-shard V2 bundle (what tf.train.Saver(write_version=V2) produces)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    tw = _TableWriter(prefix + ".index")
    tw.add(b"", _pb_int(1, 1) + _pb_int(2, 0) + _pb_bytes(3, _pb_int(1, 1)))       # num_shards 1, little endian, version{producer 1}
    off = 0
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        for name in sorted(tensors):
            a = np.asarray(tensors[name], dtype="<f4", order="C")       # (not ascontiguousarray: it turns a scalar into shape (1,))
            raw = a.tobytes()
            entry = _pb_int(1, DT_FLOAT) + _pb_bytes(2, _encode_shape(a.shape))
            if off:
                entry += _pb_int(4, off)
            entry += _pb_int(5, len(raw)) + _pb_key(6, 5) + struct.pack("<I", mask_crc(crc32c(raw)))
            tw.add(name.encode(), entry)
            data.write(raw)
            off += len(raw)
    tw.close()


# ------------------------------------------------------------------------------------------- V1 tensor-slice files
def _parse_tensor_proto(buf: bytes) -> np.ndarray:
    dtype, shape, content, floats, have_shape = 0, (), None, [], False
    for fn, wt, v in _pb_fields(buf):
        if fn == 1:
            dtype = v
        elif fn == 2:
            shape, have_shape = _parse_shape(v), True
        elif fn == 4:
            content = v
        elif fn == 5:                                 # float_val: packed (wt 2) or single fixed32 (wt 5)
            floats.append(np.frombuffer(v, dtype="<f4"))
    if dtype != DT_FLOAT:
        return None
    if content is not None:
        return np.frombuffer(content, dtype="<f4").reshape(shape).copy()
    a = np.concatenate(floats) if floats else np.zeros(0, dtype=np.float32)
    if not have_shape:
        return a
    n = int(np.prod(shape, dtype=np.int64))               # (an empty shape is a scalar: one element)
    if 0 < a.size < n:                                    # TensorProto: a short float_val repeats its last value
        a = np.concatenate([a, np.full(n - a.size, a[-1], dtype=np.float32)])
    return a.reshape(shape).copy()


def read_v1(path: str) -> Dict[str, np.ndarray]:
    """Single-file V1 checkpoint (e.g. slim's resnet_v1_50.ckpt) -> {name: fp32 array}; full slices only."""
    out = {}
    for key, val in table_entries(path):
        if key == b"":
            continue                                  # SavedTensorSliceMeta
        for fn, _, v in _pb_fields(val):              # SavedTensorSlices
            if fn != 2:
                continue
            name, tensor = None, None
            for f2, _, v2 in _pb_fields(v):           # SavedSlice{name=1, slice=2, data=3}
                if f2 == 1:
                    name = v2.decode()
                elif f2 == 3:
                    tensor = _parse_tensor_proto(v2)
            if name is not None and tensor is not None:
                if name in out:
                    raise ValueError("partitioned variable %s is not supported" % name)
                out[name] = tensor
    return out


def write_v1(path: str, tensors: Dict[str, np.ndarray]):
    """Write a V1 tensor-slice file with one full slice per tensor (test fixture / export)."""
    tw = _TableWriter(path)
    metas = b""
    for name in sorted(tensors):
        a = np.asarray(tensors[name], dtype=np.float32)
        full = b"".join(_pb_bytes(1, b"") for _ in a.shape)            # TensorSliceProto: one empty Extent per dim
        metas += _pb_bytes(1, _pb_bytes(1, name.encode()) + _pb_bytes(2, _encode_shape(a.shape)) + _pb_int(3, DT_FLOAT) +
                           _pb_bytes(4, full))
    tw.add(b"", _pb_bytes(1, metas))
    for i, name in enumerate(sorted(tensors)):
        a = np.asarray(tensors[name], dtype="<f4", order="C")
        full = b"".join(_pb_bytes(1, b"") for _ in a.shape)
        tp = _pb_int(1, DT_FLOAT) + _pb_bytes(2, _encode_shape(a.shape)) + _pb_bytes(5, a.tobytes())
        key = b"\x00" + name.encode() + b"\x00\x01" + bytes([len(a.shape)]) + b"\x00" * i      # ordered & unique; readers ignore it
        tw.add(key, _pb_bytes(2, _pb_bytes(1, name.encode()) + _pb_bytes(2, full) + _pb_bytes(3, tp)))
    tw.close()


# ------------------------------------------------------------------------------------------- front door
def _is_table(path: str) -> bool:
    """Does the file end in the SSTable footer magic (V1 checkpoint files and V2 .index files do)?"""
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            if f.tell() < 48:
                return False
            f.seek(-8, os.SEEK_END)
            return struct.unpack("<Q", f.read(8))[0] == TABLE_MAGIC
    except OSError:
        return False


def is_tf_checkpoint(path: str) -> bool:
    """A V2 prefix (<path>.index exists) or a V1 file -- recognised by the table magic, whatever it is called (DLC's resnet_v1_50.ckpt, a
    snapshot-N file without suffix); an .npz snapshot is not one."""
    return os.path.isfile(path + ".index") or (os.path.isfile(path) and _is_table(path))


def load_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """V2 prefix (has <path>.index) or V1 file -> {TF variable name: fp32 array}.  Optimiser slots
    (".../Momentum") and non-float tensors are dropped."""
    if os.path.isfile(path + ".index"):
        t = read_v2(path)
    elif os.path.isfile(path):
        t = read_v1(path)
    else:
        raise FileNotFoundError(path)
    return {k: v for k, v in t.items() if not k.endswith("/Momentum") and not k.endswith("/Adam") and not k.endswith("/Adam_1")}
