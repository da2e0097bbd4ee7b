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
# Known-answer files assembled BY HAND from the published formats by tests/_kat_ckpt.py, code that shares nothing with
# tf_checkpoint's writer or reader: own varints, own CRC-32C, own snappy emitter, explicit struct packing.  The reader must return
# the tensors these bytes describe, and the independent reader there must return what the product's writer wrote.
# ---------------------------------------------------------------------------------------------------------------------------
import _kat_ckpt as kat

_kat_crc32c, _kat_mask = kat.crc32c_bitwise, kat.mask


def _kat_bundle(tmp_path, name, tensors, snappy=False, two_blocks=False):
    prefix = str(tmp_path / name)
    first = len(sorted(tensors)[0]) + 40          # closes the first data block behind the header + first entry
    kat.bundle(prefix, tensors, compress=snappy, block_bytes=first if two_blocks else 1 << 30)
    return prefix


def test_kat_crc_and_snappy_helpers_are_sound():
    """The independent assembler's own pieces: the lane-folded CRC equals the bitwise loop, and its snappy stream (literals + overlapping
    copies) decodes to the input."""
    rng = np.random.RandomState(0)
    for n in (1 << 14, 70001, 300000):
        data = rng.randint(0, 256, size=n, dtype=np.uint8).tobytes()
        assert kat.crc32c(data) == kat.crc32c_bitwise(data)
    assert kat.crc32c(b"123456789") == 0xE3069283
    raw = np.concatenate([np.zeros(40, np.float32), np.ones(70, np.float32), rng.randn(33).astype(np.float32)]).tobytes() + b"xy"
    comp = kat.snappy(raw)
    assert len(comp) < len(raw) // 2 and tfc._snappy_decompress(comp) == raw      # runs became copies
    big = rng.randint(0, 256, size=200000, dtype=np.uint8).tobytes()
    assert tfc._snappy_decompress(kat.snappy(big)) == big


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


def _saver_like_variables(rng):
    """What tf.train.Saver() puts into a DLC / DGP snapshot when it is built after the optimiser (fit_dlc: fitdgp.py:150-152): every
    model variable, a `<var>/Momentum` slot per trainable, and the int64 `global_step`."""
    model = {"resnet_v1_50/conv1/weights": rng.randn(7, 7, 3, 64).astype(np.float32),
             "resnet_v1_50/conv1/BatchNorm/gamma": np.ones(64, np.float32),
             "resnet_v1_50/conv1/BatchNorm/beta": np.zeros(64, np.float32),
             "resnet_v1_50/conv1/BatchNorm/moving_mean": rng.randn(64).astype(np.float32),
             "resnet_v1_50/conv1/BatchNorm/moving_variance": rng.rand(64).astype(np.float32) + 0.5,
             "pose/part_pred/block4/weights": rng.randn(3, 3, 4, 8).astype(np.float32),
             "pose/part_pred/block4/biases": rng.randn(4).astype(np.float32)}
    extra = {k + "/Momentum": rng.randn(*v.shape).astype(np.float32) for k, v in model.items() if "moving_" not in k}
    extra["global_step"] = np.array(1030000, dtype=np.int64)
    return model, extra


@pytest.mark.parametrize("compress", [False, True])
def test_v2_bundle_with_momentum_slots_and_global_step(tmp_path, compress):
    """A Saver-written snapshot holds optimiser slots and an int64 global_step next to the model: they are skipped, not rejected."""
    model, extra = _saver_like_variables(np.random.RandomState(8))
    prefix = str(tmp_path / "snapshot-step0-final--0")
    n_blocks = kat.bundle(prefix, {**model, **extra}, compress=compress, block_bytes=200)
    assert n_blocks >= 3
    raw = tfc.read_v2(prefix, verify=True)
    assert "global_step" not in raw and "pose/part_pred/block4/weights/Momentum" in raw        # non-float skipped by the bundle reader
    back = weights_io.load_weights(prefix)
    assert sorted(back) == sorted(model)
    for k, v in model.items():
        assert back[k].dtype == np.float32 and np.array_equal(back[k], v), k


@pytest.mark.parametrize("compress", [True, False])
def test_v1_reader_on_hand_assembled_checkpoint(tmp_path, compress):
    """slim's ImageNet resnet_v1_50.ckpt is a V1 tensor-slice file: snappy blocks, float_val payloads, ordered-code keys, and it holds
    variables the pose network does not have (logits, mean_rgb, global_step)."""
    rng = np.random.RandomState(9)
    model, extra = _saver_like_variables(rng)
    model = {k: v for k, v in model.items() if k.startswith("resnet_v1_50")}
    model["resnet_v1_50/logits/biases"] = np.zeros(1000, np.float32)
    model["resnet_v1_50/mean_rgb"] = np.array([123.68, 116.78, 103.94], np.float32)
    path = str(tmp_path / "resnet_v1_50.ckpt")
    n_blocks = kat.v1_file(path, {**model, "global_step": np.array(7, dtype=np.int64)}, compress=compress, block_bytes=2000)
    assert n_blocks >= 3
    assert tfc.is_tf_checkpoint(path)
    back = weights_io.load_weights(path)
    assert sorted(back) == sorted(model)
    for k, v in model.items():
        assert back[k].shape == v.shape and np.array_equal(back[k], v), k
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 3] ^= 0x10                                                   # inside a data block: its checksum must notice
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        tfc.read_v1(path)


def test_product_writer_is_read_by_the_independent_reader(tmp_path):
    """The other direction: what write_v2 / weights_io.Saver put on disk, parsed by tests/_kat_ckpt.read_bundle (own table walk, own
    proto decoding, every block and tensor checksum verified with its own CRC)."""
    w = synthetic.make_weights(50, 3, True, seed=4)
    small = {k: v for k, v in w.items() if v.size <= 70000}              # the bitwise / lane CRC of the test reader: keep it to seconds
    small["pose/scalar"] = np.float32(0.25).reshape(())
    saver = weights_io.Saver(max_to_keep=2, fmt="tf")
    paths = [saver.save(small, str(tmp_path / "snapshot-step2-"), global_step=g) for g in (5, 0, 10)]
    assert paths[0].endswith("snapshot-step2--5") and paths[1].endswith("snapshot-step2--0")
    assert not os.path.exists(paths[0] + ".index") and not os.path.exists(paths[0] + ".data-00000-of-00001")      # max_to_keep
    assert weights_io.latest_checkpoint(str(tmp_path)) == paths[2]
    state = open(tmp_path / "checkpoint").read().splitlines()
    assert state == ['model_checkpoint_path: "snapshot-step2--10"', 'all_model_checkpoint_paths: "snapshot-step2--0"',
                     'all_model_checkpoint_paths: "snapshot-step2--10"']
    back = kat.read_bundle(paths[2])
    assert sorted(back) == sorted(small)
    for k, v in small.items():
        assert back[k].shape == np.asarray(v).shape and np.array_equal(back[k], v), k
