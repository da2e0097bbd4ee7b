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
