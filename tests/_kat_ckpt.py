"""TensorFlow-1 checkpoint files assembled BY HAND from the published formats -- test infrastructure that shares no code with
deepgraphpose_amd/tf_checkpoint.py (own varints, own CRC-32C, own snappy emitter, explicit struct packing), so that the product's
reader is checked against bytes it did not write and the product's writer against a reader it does not contain.

Formats restated (no TensorFlow in this image; names are files of the tensorflow r1.15 tree):
  leveldb table        core/lib/io/{table_builder,block_builder,format}.cc: data blocks of prefix-compressed entries + restart array,
                       each followed by [compression type byte][masked crc32c of block + type]; metaindex block; index block of
                       BlockHandles; 48-byte footer ending in kTableMagicNumber 0xdb4775248b80fb57
  V2 tensor bundle     core/protobuf/tensor_bundle.proto (BundleHeaderProto, BundleEntryProto), core/util/tensor_bundle/
  V1 tensor slices     core/util/saved_tensor_slice.proto (SavedTensorSlices{meta | data: SavedSlice{name, slice, data: TensorProto}}),
                       core/util/saved_tensor_slice_util.cc (EncodeTensorNameSlice: ordered-code keys), core/util/tensor_slice_writer.cc
                       (float data in TensorProto.float_val, snappy-compressed blocks)
  crc mask             core/lib/hash/crc32c.h: ((crc >> 15) | (crc << 17)) + 0xa282ead8
"""
import struct

import numpy as np

DT_FLOAT, DT_INT64 = 1, 9


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def crc32c_bitwise(data, crc=0):
    crc ^= 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)       # Castagnoli polynomial, reflected
    return crc ^ 0xFFFFFFFF


_TAB = None


def _table():
    global _TAB
    if _TAB is None:
        t = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            t[i] = c
        _TAB = t
    return _TAB


def crc32c(data):
    """CRC-32C of a byte string.  Small inputs: the bitwise loop.  Large ones (the 9-MB tensors of a ResNet): the buffer is cut into
    L lanes whose registers advance together (one numpy table look-up per byte position), then the lane registers are folded with
    the linear map "advance a register over n zero bytes" -- reg(A || B from s) = advance(reg(A from s), |B|) xor reg(B from 0)."""
    data = bytes(data)
    if len(data) < 1 << 14:
        return crc32c_bitwise(data)
    tab = _table()
    lanes = 2048
    n = len(data) // lanes
    head, tail = data[:n * lanes], data[n * lanes:]
    cols = np.frombuffer(head, dtype=np.uint8).reshape(lanes, n)
    reg = np.zeros(lanes, dtype=np.uint32)
    for j in range(n):
        reg = tab[(reg ^ cols[:, j]) & 0xFF] ^ (reg >> 8)
    # advance-by-n-zero-bytes as four 256-entry tables (it is linear: apply it to every byte value of every byte position)
    adv = []
    for pos in range(4):
        basis = (np.arange(256, dtype=np.uint32) << np.uint32(8 * pos)).astype(np.uint32)
        for _ in range(n):
            basis = tab[basis & 0xFF] ^ (basis >> 8)
        adv.append(basis)
    s = 0xFFFFFFFF
    for k in range(lanes):
        s = int(adv[0][s & 0xFF]) ^ int(adv[1][(s >> 8) & 0xFF]) ^ int(adv[2][(s >> 16) & 0xFF]) ^ int(adv[3][s >> 24]) ^ int(reg[k])
    t = int(s)
    for byte in tail:
        t = int(tab[(t ^ byte) & 0xFF]) ^ (t >> 8)
    return t ^ 0xFFFFFFFF


def mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def snappy(raw):
    """A valid snappy stream of `raw`: varint length, then literals (tags 0..59 inline length, 60 / 61: one / two length bytes) and,
    where the next 4..64 bytes repeat the 4 bytes before them (runs of one float value: zeros, ones), copies with a 2-byte offset."""
    raw = bytes(raw)
    out, pos, lit = bytearray(varint(len(raw))), 0, 0

    def flush(upto):
        nonlocal lit
        while lit < upto:
            n = min(60000, upto - lit)
            if n <= 60:
                out.append((n - 1) << 2)
            elif n <= 256:
                out.extend([60 << 2, n - 1])
            else:
                out.extend([61 << 2, (n - 1) & 0xFF, (n - 1) >> 8])
            out.extend(raw[lit:lit + n])
            lit += n

    if len(raw) > 1 << 16:            # big tensors: literals only (the run search below is a Python loop)
        flush(len(raw))
        return bytes(out)
    while pos < len(raw):
        if pos >= 4 and raw[pos:pos + 4] == raw[pos - 4:pos] and len(raw) - pos >= 4:
            run = 4
            while run < 64 and pos + run < len(raw) and raw[pos + run] == raw[pos + run - 4]:
                run += 1
            flush(pos)
            out.extend([((run - 1) << 2) | 2, 4, 0])                  # copy, 2-byte little-endian offset 4 (overlapping)
            pos += run
            lit = pos
        else:
            pos += 1
    flush(len(raw))
    return bytes(out)


def block(entries, restart_every=16, compress=False):
    """leveldb block: prefix-compressed entries, restart array, [type byte][masked crc32c of contents + type]."""
    body, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_every == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        body += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        last = k
    for r in restarts or [0]:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts) or 1)
    ctype = 0
    if compress:
        body, ctype = snappy(body), 1
    trailer = bytes([ctype]) + struct.pack("<I", mask(crc32c(bytes(body) + bytes([ctype]))))
    return bytes(body), trailer


def table(path, entries, block_bytes=4096, compress=False, restart_every=16):
    """Sorted (key, value) pairs -> one table file; a data block is closed once it holds >= block_bytes.  Returns the block count."""
    out, index, cur, size = bytearray(), [], [], 0

    def close():
        nonlocal cur, size
        if cur:
            body, trailer = block(cur, restart_every=restart_every, compress=compress)
            index.append((cur[-1][0] + b"\x00", len(out), len(body)))            # any separator key >= the block's last key
            out.extend(body + trailer)
            cur, size = [], 0

    for k, v in entries:
        cur.append((k, v))
        size += len(k) + len(v)
        if size >= block_bytes:
            close()
    close()
    mbody, mtrailer = block([])
    moff = len(out)
    out += mbody + mtrailer
    ibody, itrailer = block([(k, varint(o) + varint(s)) for k, o, s in index])
    ioff = len(out)
    out += ibody + itrailer
    footer = varint(moff) + varint(len(mbody)) + varint(ioff) + varint(len(ibody))
    footer += b"\x00" * (40 - len(footer)) + bytes.fromhex("57fb808b247547db")          # kTableMagicNumber, little endian
    out += footer
    with open(path, "wb") as f:
        f.write(bytes(out))
    return len(index)


def _shape_proto(shape):
    return b"".join(bytes([0x12]) + varint(len(d)) + d for d in (bytes([0x08]) + varint(int(s)) for s in shape))


def _as_le(a):
    a = np.asarray(a)
    if a.dtype == np.int64:
        return a.astype("<i8"), DT_INT64
    return a.astype("<f4"), DT_FLOAT


def bundle(prefix, tensors, compress=False, block_bytes=4096, restart_every=2):
    """Write <prefix>.index / <prefix>.data-00000-of-00001 for {variable: float32 (or int64: global_step) array}."""
    data, entries = bytearray(), []
    header = bytes([0x08, 0x01, 0x10, 0x00, 0x1A, 0x02, 0x08, 0x01])      # BundleHeaderProto{num_shards: 1, LITTLE, version{producer: 1}}
    entries.append((b"", header))
    for key in sorted(tensors):
        a, dt = _as_le(tensors[key])                                      # (ascontiguousarray would turn a scalar into shape (1,))
        raw = a.tobytes()
        shape = _shape_proto(a.shape)
        val = bytes([0x08, dt])                                           # dtype
        val += bytes([0x12]) + varint(len(shape)) + shape                 # shape: TensorShapeProto{dim{size}...}
        if len(data):
            val += bytes([0x20]) + varint(len(data))                      # offset (shard_id 0 and offset 0 are proto defaults: omitted)
        val += bytes([0x28]) + varint(len(raw))                           # size
        val += bytes([0x35]) + struct.pack("<I", mask(crc32c(raw)))       # crc32c: fixed32
        entries.append((key.encode(), val))
        data += raw
    n_blocks = table(prefix + ".index", entries, block_bytes=block_bytes, compress=compress, restart_every=restart_every)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    return n_blocks


def _ordered_string(s):
    return s.replace(b"\x00", b"\x00\xff").replace(b"\xff", b"\xff\x00") + b"\x00\x01"


def v1_file(path, tensors, compress=True, block_bytes=4096):
    """Single-file V1 checkpoint (the format of slim's resnet_v1_50.ckpt): key "" -> SavedTensorSlices{meta}, then one
    SavedTensorSlices{data} per tensor under its ordered-code key; float data in TensorProto.float_val (packed), int64 in int64_val."""
    metas, entries = b"", []
    for name in sorted(tensors):
        a, dt = _as_le(tensors[name])
        nd = a.ndim
        full = b"".join(bytes([0x0A, 0x00]) for _ in range(nd))                       # TensorSliceProto: one empty Extent per dim (full)
        shape = _shape_proto(a.shape)
        nm = name.encode()
        meta = bytes([0x0A]) + varint(len(nm)) + nm + bytes([0x12]) + varint(len(shape)) + shape + bytes([0x18, dt])
        meta += bytes([0x22]) + varint(len(full)) + full                              # SavedSliceMeta{name, shape, type, slice}
        metas += bytes([0x0A]) + varint(len(meta)) + meta
        if dt == DT_FLOAT:
            payload = bytes([0x2A]) + varint(a.nbytes) + a.tobytes()                  # float_val = 5, packed
        else:
            packed = b"".join(varint(int(v) & 0xFFFFFFFFFFFFFFFF) for v in a.reshape(-1))
            payload = bytes([0x52]) + varint(len(packed)) + packed                    # int64_val = 10, packed
        tp = bytes([0x08, dt, 0x12]) + varint(len(shape)) + shape + payload           # TensorProto{dtype, tensor_shape, *_val}
        sl = bytes([0x0A]) + varint(len(nm)) + nm + bytes([0x12]) + varint(len(full)) + full + bytes([0x1A]) + varint(len(tp)) + tp
        # EncodeTensorNameSlice: WriteNumIncreasing(0) | WriteString(name) | WriteNumIncreasing(dims) | per dim start 0, length -1
        key = b"\x00" + _ordered_string(nm) + (b"\x01" + bytes([nd]) if nd else b"\x00") + b"\x80\x7f" * nd
        entries.append((key, bytes([0x12]) + varint(len(sl)) + sl))                   # SavedTensorSlices{data = 2}
    versions = bytes([0x08, 0x01])                                                    # VersionDef{producer: 1}
    meta_msg = metas + bytes([0x12]) + varint(len(versions)) + versions
    entries.sort(key=lambda kv: kv[0])
    entries.insert(0, (b"", bytes([0x0A]) + varint(len(meta_msg)) + meta_msg))        # SavedTensorSlices{meta = 1}
    return table(path, entries, block_bytes=block_bytes, compress=compress)


# ------------------------------------------------------------------------------------------------- independent READER (V2 bundles)
def _get_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if b < 0x80:
            return val, pos
        shift += 7


def _read_block(raw, off, size):
    body, ctype = raw[off:off + size], raw[off + size]
    (crc,) = struct.unpack("<I", raw[off + size + 1:off + size + 5])
    assert mask(crc32c(body + bytes([ctype]))) == crc, "block checksum"
    assert ctype == 0, "this reader takes uncompressed blocks (what the product's writer emits)"
    (nres,) = struct.unpack("<I", body[-4:])
    end, pos, key, out = len(body) - 4 - 4 * nres, 0, b"", []
    while pos < end:
        shared, pos = _get_varint(body, pos)
        non_shared, pos = _get_varint(body, pos)
        vlen, pos = _get_varint(body, pos)
        key = key[:shared] + body[pos:pos + non_shared]
        pos += non_shared
        out.append((key, body[pos:pos + vlen]))
        pos += vlen
    return out


def read_bundle(prefix):
    """<prefix>.index + .data-00000-of-00001 -> {name: array}, checking every checksum, by this file's rules alone."""
    raw = open(prefix + ".index", "rb").read()
    assert raw[-8:] == bytes.fromhex("57fb808b247547db"), "table magic"
    footer = raw[-48:]
    _, pos = _get_varint(footer, 0)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isz, pos = _get_varint(footer, pos)
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    out, seen_header = {}, False
    for _, handle in _read_block(raw, ioff, isz):
        boff, p2 = _get_varint(handle, 0)
        bsz, _ = _get_varint(handle, p2)
        for key, val in _read_block(raw, boff, bsz):
            if key == b"":
                seen_header = True
                assert val[:4] == bytes([0x08, 0x01, 0x10, 0x00]), "BundleHeaderProto{num_shards 1, little endian}"
                continue
            f, pos = {"dtype": 0, "shape": [], "offset": 0, "size": 0, "crc": None}, 0
            while pos < len(val):
                tag, pos = _get_varint(val, pos)
                fn, wt = tag >> 3, tag & 7
                if wt == 0:
                    v, pos = _get_varint(val, pos)
                elif wt == 2:
                    n, pos = _get_varint(val, pos)
                    v, pos = val[pos:pos + n], pos + n
                else:
                    assert wt == 5
                    v, pos = struct.unpack("<I", val[pos:pos + 4])[0], pos + 4
                if fn == 1:
                    f["dtype"] = v
                elif fn == 2:
                    q = 0
                    while q < len(v):
                        assert v[q] == 0x12
                        n, q = _get_varint(v, q + 1)
                        dim = v[q:q + n]
                        q += n
                        f["shape"].append(_get_varint(dim, 1)[0] if dim else 0)
                elif fn == 3:
                    assert v == 0, "single shard"
                elif fn == 4:
                    f["offset"] = v
                elif fn == 5:
                    f["size"] = v
                elif fn == 6:
                    f["crc"] = v
            chunk = data[f["offset"]:f["offset"] + f["size"]]
            assert len(chunk) == f["size"] and mask(crc32c(chunk)) == f["crc"], "tensor checksum of %s" % key.decode()
            out[key.decode()] = np.frombuffer(chunk, dtype="<f4" if f["dtype"] == DT_FLOAT else "<i8").reshape(f["shape"]).copy()
    assert seen_header
    return out
