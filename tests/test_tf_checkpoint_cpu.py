"""TF-1 checkpoint reader/writer: round trips + format invariants.  No TensorFlow here, so
these pin the reader to the writer and to the published constants, not to TF-produced files."""
import os
import struct

import numpy as np
import pytest

from deepgraphpose_amd import synthetic, tf_checkpoint as tfc, weights_io


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert tfc.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert tfc.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfc.crc32c(b"123456789") == 0xE3069283


def test_crc32c_c_helper_matches_python(lib_built):
    data = np.random.RandomState(1).randint(0, 256, size=10007, dtype=np.uint8).tobytes()
    slow = tfc.crc32c(data[:4000])                                           # < 4096: pure-python path
    for b in data[4000:]:
        slow = tfc._CRC[(slow ^ 0xFFFFFFFF ^ b) & 0xFF] ^ ((slow ^ 0xFFFFFFFF) >> 8) ^ 0xFFFFFFFF
    assert tfc.crc32c(data) == slow


def test_snappy_decoder():
    # literal "abcd" + copy(offset 4, len 8) -> "abcdabcdabcd"; varint length 12
    blob = bytes([12, (4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4])
    assert tfc._snappy_decompress(blob) == b"abcdabcdabcd"


def test_v2_round_trip_full_network(tmp_path):
    w = synthetic.make_weights(50, 4, True, seed=3)
    prefix = str(tmp_path / "snapshot-step2-final--0")
    weights_io.save_weights(prefix, w, fmt="tf")
    assert os.path.isfile(prefix + ".index") and os.path.isfile(prefix + ".data-00000-of-00001")
    with open(prefix + ".index", "rb") as f:
        f.seek(-8, os.SEEK_END)
        assert struct.unpack("<Q", f.read(8))[0] == 0xDB4775248B80FB57
    back = weights_io.load_weights(prefix)
    assert sorted(back) == sorted(w)
    for k in w:
        assert back[k].shape == w[k].shape and np.array_equal(back[k], w[k]), k
    assert tfc.read_v2(prefix, verify=True).keys() == back.keys()


def test_v2_detects_corruption(tmp_path):
    prefix = str(tmp_path / "m")
    tfc.write_v2(prefix, {"a/weights": np.arange(6, dtype=np.float32).reshape(2, 3)})
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[3] ^= 0x40
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        tfc.read_v2(prefix)


def test_v1_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    t = {"resnet_v1_50/conv1/weights": rng.randn(7, 7, 3, 64).astype(np.float32),
         "resnet_v1_50/conv1/BatchNorm/gamma": rng.randn(64).astype(np.float32),
         "resnet_v1_50/conv1/BatchNorm/gamma/Momentum": rng.randn(64).astype(np.float32),
         "resnet_v1_50/logits/biases": rng.randn(1000).astype(np.float32)}
    path = str(tmp_path / "resnet_v1_50.ckpt")
    tfc.write_v1(path, t)
    back = weights_io.load_weights(path)
    assert "resnet_v1_50/conv1/BatchNorm/gamma/Momentum" not in back          # optimiser slots dropped
    for k in back:
        assert np.array_equal(back[k], t[k])
    assert len(back) == 3


def test_not_a_checkpoint(tmp_path):
    p = tmp_path / "junk.ckpt"
    p.write_bytes(b"x" * 100)
    with pytest.raises(Exception):
        weights_io.load_weights(str(p))


# ---------------------------------------------------------------------------------------------------------------------------
# Known-answer files assembled BY HAND from the published formats (leveldb table_format.md; tensorflow/core/protobuf/
# tensor_bundle.proto; tensorflow/core/lib/hash/crc32c.h for the mask) with code that shares nothing with tf_checkpoint's writer:
# own varints, own bitwise CRC-32C, explicit struct packing.  The reader must return the tensors these bytes describe.
# ---------------------------------------------------------------------------------------------------------------------------
def _kat_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _kat_crc32c(data):
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)       # Castagnoli polynomial, reflected
    return crc ^ 0xFFFFFFFF


def _kat_mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _kat_block(entries, restart_every=16, snappy=False):
    """leveldb block: prefix-compressed entries, restart array, [type byte][masked crc32c of contents + type]."""
    body, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_every == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        body += _kat_varint(shared) + _kat_varint(len(k) - shared) + _kat_varint(len(v)) + k[shared:] + v
        last = k
    for r in restarts or [0]:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts) or 1)
    ctype = 0
    if snappy:      # snappy stream of literals only (tag = (len - 1) << 2 for len <= 60, else tag 60 << 2 + one length byte)
        raw, comp, pos = bytes(body), bytearray(_kat_varint(len(body))), 0
        while pos < len(raw):
            n = min(200, len(raw) - pos)
            comp += bytes([(n - 1) << 2]) if n <= 60 else bytes([60 << 2, n - 1])
            comp += raw[pos:pos + n]
            pos += n
        body, ctype = comp, 1
    trailer = bytes([ctype]) + struct.pack("<I", _kat_mask(_kat_crc32c(bytes(body) + bytes([ctype]))))
    return bytes(body), trailer


def _kat_bundle(tmp_path, name, tensors, snappy=False, two_blocks=False):
    """Write <name>.index / <name>.data-00000-of-00001 for {variable: float32 array} without tf_checkpoint's writer."""
    data, entries = bytearray(), []
    header = bytes([0x08, 0x01, 0x10, 0x00, 0x1A, 0x02, 0x08, 0x01])      # BundleHeaderProto{num_shards: 1, LITTLE, version{producer: 1}}
    entries.append((b"", header))
    for key in sorted(tensors):
        a = np.asarray(tensors[key], dtype="<f4")                     # (ascontiguousarray would turn a scalar into shape (1,))
        raw = a.tobytes()
        shape = b"".join(bytes([0x12]) + _kat_varint(len(d)) + d for d in (bytes([0x08]) + _kat_varint(s) for s in a.shape))
        val = bytes([0x08, 0x01])                                         # dtype: DT_FLOAT
        val += bytes([0x12]) + _kat_varint(len(shape)) + shape            # shape: TensorShapeProto{dim{size}...}
        if len(data):
            val += bytes([0x20]) + _kat_varint(len(data))                 # offset (shard_id 0 and offset 0 are proto defaults: omitted)
        val += bytes([0x28]) + _kat_varint(len(raw))                      # size
        val += bytes([0x35]) + struct.pack("<I", _kat_mask(_kat_crc32c(raw)))     # crc32c: fixed32
        entries.append((key.encode(), val))
        data += raw
    groups = [entries[:2], entries[2:]] if two_blocks and len(entries) > 2 else [entries]
    out, index = bytearray(), []
    for g in groups:
        body, trailer = _kat_block(g, restart_every=2, snappy=snappy)
        index.append((g[-1][0] + b"\x00", len(out), len(body)))            # any separator key >= the block's last key
        out += body + trailer
    mbody, mtrailer = _kat_block([])
    moff = len(out)
    out += mbody + mtrailer
    ibody, itrailer = _kat_block([(k, _kat_varint(o) + _kat_varint(s)) for k, o, s in index])
    ioff = len(out)
    out += ibody + itrailer
    footer = _kat_varint(moff) + _kat_varint(len(mbody)) + _kat_varint(ioff) + _kat_varint(len(ibody))
    footer += b"\x00" * (40 - len(footer)) + bytes.fromhex("57fb808b247547db")          # kTableMagicNumber, little endian
    out += footer
    prefix = str(tmp_path / name)
    open(prefix + ".index", "wb").write(bytes(out))
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    return prefix


@pytest.mark.parametrize("snappy,two_blocks", [(False, False), (True, False), (False, True)])
def test_v2_reader_on_hand_assembled_bundle(tmp_path, snappy, two_blocks):
    rng = np.random.RandomState(5)
    tensors = {"pose/part_pred/block4/biases": rng.randn(4).astype(np.float32),
               "pose/part_pred/block4/weights": rng.randn(3, 3, 4, 8).astype(np.float32),
               "resnet_v1_50/conv1/BatchNorm/gamma": rng.randn(64).astype(np.float32),
               "resnet_v1_50/conv1/weights": rng.randn(7, 7, 3, 64).astype(np.float32),
               "scalar": np.float32(2.5).reshape(())}
    prefix = _kat_bundle(tmp_path, "kat", tensors, snappy=snappy, two_blocks=two_blocks)
    assert _kat_crc32c(b"123456789") == 0xE3069283 and tfc.crc32c(b"123456789") == 0xE3069283
    assert tfc.is_tf_checkpoint(prefix)
    back = tfc.read_v2(prefix, verify=True)
    assert sorted(back) == sorted(tensors)
    for k, v in tensors.items():
        assert back[k].shape == v.shape and back[k].dtype == np.float32 and np.array_equal(back[k], v), k
    assert sorted(weights_io.load_weights(prefix)) == sorted(tensors)
    # and the writer's output read back by the independent assembly's rules: same footer magic, same entry bytes for one tensor
    w2 = str(tmp_path / "w2")
    tfc.write_v2(w2, {"scalar": tensors["scalar"]})
    ent = dict(tfc.table_entries(w2 + ".index"))
    assert ent[b"scalar"][:2] == bytes([0x08, 0x01]) and ent[b"scalar"][-5] == 0x35
    assert struct.unpack("<I", ent[b"scalar"][-4:])[0] == _kat_mask(_kat_crc32c(tensors["scalar"].tobytes()))


def test_v2_reader_rejects_truncated_and_corrupt_files(tmp_path):
    """Every single-byte corruption of the index is either detected (ValueError & co.) or harmless; nothing hangs, crashes or
    returns a tensor of the wrong shape; truncation anywhere is detected."""
    rng = np.random.RandomState(6)
    tensors = {"a/w": rng.randn(2, 3).astype(np.float32), "a/b": rng.randn(3).astype(np.float32)}
    prefix = _kat_bundle(tmp_path, "fz", tensors)
    good = open(prefix + ".index", "rb").read()
    bad_exc = (ValueError, IndexError, KeyError, struct.error, OverflowError, EOFError, UnicodeDecodeError, OSError, MemoryError)
    detected = 0
    for pos in range(len(good)):
        mutated = bytearray(good)
        mutated[pos] ^= 0x5A
        open(prefix + ".index", "wb").write(bytes(mutated))
        try:
            back = tfc.read_v2(prefix, verify=True)
        except bad_exc:
            detected += 1
            continue
        for k, v in back.items():                       # undetected only where the byte is not covered by a checksum (footer padding)
            assert k in tensors and np.array_equal(v, tensors[k])
    # not covered by any checksum the reader consults: the footer's zero padding (~34 bytes) and the unused metaindex block (8 + 5)
    assert detected >= len(good) - 52
    for cut in (0, 10, len(good) // 2, len(good) - 49, len(good) - 8, len(good) - 1):
        open(prefix + ".index", "wb").write(good[:cut])
        with pytest.raises(bad_exc):
            tfc.read_v2(prefix, verify=True)
    open(prefix + ".index", "wb").write(good)
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    open(prefix + ".data-00000-of-00001", "wb").write(data[:-3])            # short data shard
    with pytest.raises(bad_exc):
        tfc.read_v2(prefix, verify=True)
    flipped = bytearray(data)
    flipped[5] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(flipped))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read_v2(prefix, verify=True)


def test_scalars_keep_rank_zero_and_v1_files_are_recognised_by_content(tmp_path):
    """Found by scripts/fuzz_tf_checkpoint.py: both writers turned a rank-0 variable into shape (1,) (np.ascontiguousarray promotes 0-d arrays), the V1
    parser returned the flat array for an EMPTY shape field (a scalar) as if the field were missing, and a V1 file was only recognised when its name
    ended in .ckpt.  Also TensorProto's rule that a short float_val repeats its last value."""
    from deepgraphpose_amd import tf_checkpoint as tfc
    tensors = {"a/scalar": np.float32(2.5), "a/vec": np.arange(3, dtype=np.float32), "a/empty": np.zeros((0, 4), np.float32)}
    v2 = str(tmp_path / "model.ckpt-7")
    tfc.write_v2(v2, tensors)
    v1 = str(tmp_path / "snapshot_without_suffix")
    tfc.write_v1(v1, tensors)
    for got in (tfc.read_v2(v2, verify=True), tfc.read_v1(v1), tfc.load_checkpoint(v1)):
        assert got["a/scalar"].shape == () and float(got["a/scalar"]) == 2.5
        assert got["a/vec"].shape == (3,) and got["a/empty"].shape == (0, 4)
    assert tfc.is_tf_checkpoint(v1) and tfc.is_tf_checkpoint(v2)
    npz = str(tmp_path / "weights.npz")
    np.savez(npz, **{k.replace("/", "__"): v for k, v in tensors.items()})
    assert not tfc.is_tf_checkpoint(npz) and not tfc.is_tf_checkpoint(str(tmp_path / "missing"))
    # a TensorProto whose float_val holds ONE value for a [2, 3] tensor (TF stores constant tensors that way)
    proto = tfc._pb_int(1, tfc.DT_FLOAT) + tfc._pb_bytes(2, tfc._encode_shape((2, 3))) + tfc._pb_bytes(5, np.float32(7.0).tobytes())
    t = tfc._parse_tensor_proto(proto)
    assert t.shape == (2, 3) and (t == 7.0).all()
