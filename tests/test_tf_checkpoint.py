"""TF-1 tensor-bundle reader/writer (hpmn_amd/tf_checkpoint.py): format primitives against known answers, round
trips through multi-block tables, corruption is detected, and the TF 1.4 variable names."""
import os
import struct

import math

import numpy as np
import pytest

from hpmn_amd import tf_checkpoint as T


def test_crc32c_known_answers_and_masking():
    assert T.crc32c(b"123456789") == 0xE3069283                       # the standard CRC-32C check value
    assert T.crc32c(b"\x00" * 32) == 0x8A9136AA                       # RFC 3720 B.4 test vectors
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"6789", T.crc32c(b"12345")) == T.crc32c(b"123456789")      # extend
    a = np.arange(1000, dtype=np.float32)
    assert T.crc32c(a) == T.crc32c(a.tobytes())
    assert T.mask_crc(0) == 0xA282EAD8 and T.mask_crc(0xE3069283) == ((0xE3069283 >> 15 | 0xE3069283 << 17) + 0xA282EAD8) & 0xFFFFFFFF


def test_slow_crc_fallback_agrees(monkeypatch):
    want = T.crc32c(b"hierarchical periodic memory")
    monkeypatch.setattr(T, "_host", False)
    assert T.crc32c(b"hierarchical periodic memory") == want


def test_table_round_trip_multi_block_prefix_compression(tmp_path):
    items = [(b"", b"header")] + [(("User/dense_%03d/kernel" % i).encode(), os.urandom(40 + i % 7)) for i in range(500)]
    items.sort()
    p = str(tmp_path / "t.index")
    T.write_table(p, items, block_size=512)
    assert T.read_table(p) == items
    raw = open(p, "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57 and len(raw) < sum(len(k) + len(v) for k, v in items) + 4000
    # a flipped byte in a data block is caught by the block checksum
    bad = bytearray(raw)
    bad[100] ^= 1
    open(p, "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="checksum"):
        T.read_table(p)


def test_bundle_round_trip_scalars_shapes_dtypes_and_corruption(tmp_path):
    rng = np.random.default_rng(0)
    t = {"Embedding/emb_mtx": rng.normal(size=(1000, 16)).astype(np.float32),
         "output/beta1_power": np.asarray(0.81, np.float32), "step": np.asarray(7, np.int64),
         "empty": np.zeros((0, 4), np.float32), "ids": np.arange(12, dtype=np.int32).reshape(3, 4)}
    for i in range(200):
        t["User/dense_%d/bias" % i] = rng.normal(size=(i % 5 + 1,)).astype(np.float32)
    prefix = str(tmp_path / "ckpt" / "model.ckpt")
    T.write_bundle(prefix, t)
    assert sorted(os.listdir(str(tmp_path / "ckpt"))) == ["checkpoint", "model.ckpt.data-00000-of-00001", "model.ckpt.index"]
    assert open(str(tmp_path / "ckpt" / "checkpoint")).read().startswith('model_checkpoint_path: "model.ckpt"')
    r = T.read_bundle(prefix)
    assert set(r) == set(t)
    for k in t:
        assert r[k].shape == t[k].shape and r[k].dtype == t[k].dtype and np.array_equal(r[k], t[k]), k
    assert set(T.read_bundle(prefix, names=["step"])) == {"step"}
    # the data file is the tensors back to back in key order; the header entry is the bundle header proto
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(v.nbytes for v in t.values())
    assert T.read_table(prefix + ".index")[0] == (b"", T.HEADER_PROTO)
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[10] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        T.read_bundle(prefix)


def test_entry_proto_bytes_are_the_documented_wire_format():
    # BundleEntryProto{dtype: DT_FLOAT, shape{dim{size:3} dim{size:16}}, offset: 300, size: 192, crc32c: 0x01020304}
    got = T._encode_entry(1, (3, 16), 300, 192, 0x01020304)
    assert got == bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x03, 0x12, 0x02, 0x08, 0x10,
                         0x20, 0xAC, 0x02, 0x28, 0xC0, 0x01, 0x35, 0x04, 0x03, 0x02, 0x01])
    e = T._decode_entry(got)
    assert (e["dtype"], e["shape"], e["offset"], e["size"], e["crc32c"]) == (1, (3, 16), 300, 192, 0x01020304)


def test_tf14_variable_names_and_model_export_import(tmp_path):
    names = ["Embedding/emb_mtx", "User/GRU0/gates/kernel", "User/GRU0/gates/bias", "User/GRU1/candidate/kernel",
             "User/dense/kernel", "User/map", "User/dense_3/bias", "output/bn1/gamma", "output/fc1/kernel"]
    m = T.tf_name_map(names)
    assert m["User/GRU0/gates/kernel"] == "User/GRU0/rnn/gru_cell/gates/kernel"       # code/hpmn.py:117-120
    assert m["User/GRU1/candidate/kernel"] == "User/GRU1/rnn/gru_cell/candidate/kernel"
    assert all(m[k] == k for k in names if "GRU" not in k)
    rng = np.random.default_rng(1)
    shapes = {"Embedding/emb_mtx": (50, 16), "User/GRU0/gates/kernel": (80, 64), "User/GRU0/gates/bias": (64,),
              "output/bn1/gamma": (80,), "output/fc3/bias": (1,)}
    p = {k: rng.normal(size=s).astype(np.float32) for k, s in shapes.items()}
    am = {k: rng.normal(size=s).astype(np.float32) for k, s in shapes.items()}
    av = {k: rng.random(size=s).astype(np.float32) for k, s in shapes.items()}
    prefix = str(tmp_path / "model.ckpt")
    T.export_model(prefix, p, am, av, adam_t=37, mask_table_rows=50)
    have = T.read_bundle(prefix)
    assert "User/GRU0/rnn/gru_cell/gates/kernel" in have and "output/Embedding/emb_mtx/Adam_1" in have
    assert have["Embedding/mask_lookup_table"].shape == (50, 1) and have["Embedding/mask_lookup_table"][0, 0] == 0
    assert np.all(have["output/bn1/moving_variance"] == 1) and np.all(have["output/bn1/moving_mean"] == 0)
    np.testing.assert_allclose(float(have["output/beta1_power"]), 0.9 ** 38, rtol=1e-6)
    p2, m2, v2, t2 = T.import_model(prefix, shapes)
    assert t2 == 37 and all(np.array_equal(p2[k], p[k]) and np.array_equal(m2[k], am[k]) and np.array_equal(v2[k], av[k])
                            for k in shapes)
    # a weights-only checkpoint (e.g. trainable variables saved elsewhere) restores with fresh Adam state
    T.write_bundle(prefix, {T.tf_name_map([k])[k]: v for k, v in p.items()})
    p3, m3, v3, t3 = T.import_model(prefix, shapes)
    assert m3 is None and t3 == 0 and np.array_equal(p3["User/GRU0/gates/bias"], p["User/GRU0/gates/bias"])
    with pytest.raises(ValueError, match="shape"):
        T.import_model(prefix, dict(shapes, **{"output/fc3/bias": (2,)}))
    with pytest.raises(KeyError):
        T.import_model(prefix, dict(shapes, **{"User/map": (64, 64)}))


@pytest.mark.parametrize("t", [0, 7, 970, 1000, 5000, 200000])
def test_adam_step_count_survives_the_round_trip(tmp_path, t):
    """float32 beta1^(t+1) underflows near t = 985: the step count comes back from the exact tensor our writer adds, and
    for a bundle without it (one TF wrote) from beta2_power, which resolves it well past 5000 steps."""
    from hpmn_amd import tf_checkpoint as tfc
    shapes = {"Embedding/emb_mtx": (5, 4)}
    p = {"Embedding/emb_mtx": np.arange(20, dtype=np.float32).reshape(5, 4)}
    z = {"Embedding/emb_mtx": np.zeros((5, 4), np.float32)}
    prefix = str(tmp_path / "m.ckpt")
    tfc.export_model(prefix, p, z, z, t, 0.9, 0.999)
    assert tfc.import_model(prefix, shapes)[3] == t
    have = tfc.read_bundle(prefix)
    del have[tfc.STEP_TENSOR]                                     # what a TF-written checkpoint looks like
    tfc.write_bundle(prefix, have)
    got = tfc.import_model(prefix, shapes)[3]
    if t <= 5000:
        assert got == t
    else:
        # both powers have underflowed (float32 0.999^200001 == 0): the count saturates where TF's own bias correction has
        # become 1 -- NOT 0, which would restart the correction (lr_t = 0.32 lr) on late-step moments
        assert got >= 80000
        lr_t = math.sqrt(1.0 - 0.999 ** got) / (1.0 - 0.9 ** got)
        assert abs(lr_t - 1.0) < 1e-6


def test_reader_against_a_bundle_assembled_by_hand_from_the_published_format(tmp_path):
    """The reader is otherwise only ever checked against our own writer.  This bundle is put together here, byte by
    byte, from the published formats (LevelDB table format doc: blocks of prefix-compressed entries + restart array +
    1-byte type + masked crc32c trailer, 48-byte footer; tensorflow/core/protobuf/tensor_bundle.proto and
    tensor_shape.proto for the values), with a bitwise crc32c that shares nothing with the product's table-driven C
    helper: two tensors, one data shard."""
    import struct
    from hpmn_amd import tf_checkpoint as tfc

    def crc32c(data):                                   # Castagnoli, reflected, bit by bit (RFC 3720 appendix B.4)
        crc = 0xFFFFFFFF
        for byte in data:
            crc ^= byte
            for _ in range(8):
                crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
        return crc ^ 0xFFFFFFFF
    assert crc32c(b"123456789") == 0xE3069283           # the check value of the CRC catalogue

    def masked(crc):                                    # leveldb/util/crc32c.h Mask()
        return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF

    a = struct.pack("<2f", 1.0, 2.0)                    # tensor "a": float32 [2]
    bc = struct.pack("<2i", 7, -3)                      # tensor "b/c": int32 [1, 2]
    data = a + bc
    header = bytes([0x08, 0x01,                         # BundleHeaderProto.num_shards = 1
                    0x1A, 0x02, 0x08, 0x01])            # .version { producer: 1 }
    entry_a = (bytes([0x08, 0x01,                       # dtype = DT_FLOAT
                      0x12, 0x04, 0x12, 0x02, 0x08, 0x02,   # shape { dim { size: 2 } }
                      0x28, 0x08,                       # size = 8   (shard_id 0 and offset 0 are defaults: absent)
                      0x35]) + struct.pack("<I", masked(crc32c(a))))
    entry_bc = (bytes([0x08, 0x03,                      # dtype = DT_INT32
                       0x12, 0x08, 0x12, 0x02, 0x08, 0x01, 0x12, 0x02, 0x08, 0x02,   # shape { dim {1} dim {2} }
                       0x20, 0x08,                      # offset = 8
                       0x28, 0x08,                      # size = 8
                       0x35]) + struct.pack("<I", masked(crc32c(bc))))

    def block(entries):                                 # one restart point, no shared prefixes
        body = b""
        for key, value in entries:
            assert len(key) < 128 and len(value) < 128  # (single-byte varints)
            body += bytes([0, len(key), len(value)]) + key + value
        body += struct.pack("<II", 0, 1)                # restart[0] = 0, num_restarts = 1
        return body

    def with_trailer(body):
        return body + b"\x00" + struct.pack("<I", masked(crc32c(body + b"\x00")))       # type 0 = uncompressed

    data_block = block([(b"", header), (b"a", entry_a), (b"b/c", entry_bc)])
    meta_block = block([])
    index_block = block([(b"c", bytes([0, len(data_block)]))])      # key >= last key of the data block -> its handle
    off_meta = len(data_block) + 5
    off_index = off_meta + len(meta_block) + 5
    footer = bytes([off_meta, len(meta_block), off_index, len(index_block)])
    footer = footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    table = with_trailer(data_block) + with_trailer(meta_block) + with_trailer(index_block) + footer
    assert max(off_meta, off_index, len(data_block), len(index_block)) < 128 and len(footer) == 48

    prefix = str(tmp_path / "hand.ckpt")
    with open(prefix + ".index", "wb") as f:
        f.write(table)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(data)
    got = tfc.read_bundle(prefix)
    assert sorted(got) == ["a", "b/c"]
    assert got["a"].dtype == np.float32 and got["a"].tolist() == [1.0, 2.0]
    assert got["b/c"].dtype == np.int32 and got["b/c"].shape == (1, 2) and got["b/c"].tolist() == [[7, -3]]
    # and the writer produces a file this independent description agrees with, entry by entry
    tfc.write_bundle(str(tmp_path / "ours.ckpt"), {"a": got["a"], "b/c": got["b/c"]})
    ours = open(str(tmp_path / "ours.ckpt") + ".index", "rb").read()
    assert entry_a in ours and entry_bc in ours and ours[-8:] == table[-8:]
    assert open(str(tmp_path / "ours.ckpt") + ".data-00000-of-00001", "rb").read() == data
