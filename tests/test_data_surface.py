"""Input-format surface: our DataLoader / front_padding / pickle reader against fixtures produced
by executing the reference's own DataLoader and front_padding (tests/golden/make_golden.py)."""
import json
import os
import pickle

import numpy as np

from hpmn_amd import datasets
from hpmn_amd.data_loader import DataLoader, DataLoader_Mul, parse_xlong_line

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _fixture():
    with open(os.path.join(GOLD, "input_surface.json")) as f:
        return json.load(f)


def test_front_padding_matches_reference_output():
    fx = _fixture()
    for raw, want in zip(fx["raw"], fx["front_padded"]):
        got = datasets.front_padding(raw, fx["user_max"], fx["user_dim"], fx["item_max"], fx["item_dim"])
        assert list(got) == want


def test_dataloader_matches_reference_batches():
    fx = _fixture()
    data = [tuple(s) for s in fx["front_padded"]]
    for case in fx["loader"]:
        got = list(DataLoader(data, case["batch_size"]))
        assert len(got) == len(case["batches"])
        for (i, (label, ipart, ilen, upart, ulen)), want in zip(got, case["batches"]):
            assert i == want["i"]                      # 1-based batch index
            assert label == want["label"] and ilen == want["item_part_len"] and ulen == want["user_part_len"]
            assert isinstance(ipart, np.ndarray) and isinstance(upart, np.ndarray)
            np.testing.assert_array_equal(ipart, np.asarray(want["item_part"]))
            np.testing.assert_array_equal(upart, np.asarray(want["user_part"]))


def test_dataloader_empty_and_exact_multiple():
    assert list(DataLoader([], 4)) == []
    data = [tuple(s) for s in _fixture()["front_padded"]][:6]
    assert [len(b[1][0]) for b in DataLoader(data, 3)] == [3, 3]


def test_pickle_roundtrip_three_consecutive_dumps(tmp_path):
    tr, te, fs = datasets.make_synthetic_amazon(n_samples=20, n_item=50, n_cate=5, n_user=30, max_len=12,
                                                item_max=4, seed=1)
    for proto in (0, 2):          # protocol 0 is what the Python-2 reference writes
        path = str(tmp_path / ("d%d.pkl" % proto))
        datasets.save_dataset_pkl(path, tr, te, fs, protocol=proto)
        a, b, c = datasets.load_dataset_pkl(path)
        assert a == tr and b == te and c == fs


def test_reads_python2_style_protocol0_stream(tmp_path):
    # a hand-written protocol-0 pickle stream (text opcodes, as cPickle.dump writes by default)
    sample = (1, [[0, 0, 0], [7, 8, 9]], 1, [[0, 0], [8, 7]], 1)
    path = str(tmp_path / "py2.pkl")
    with open(path, "wb") as f:
        for obj in ([sample], [sample, sample], 123):
            f.write(pickle.dumps(obj, protocol=0))
    tr, te, fs = datasets.load_dataset_pkl(path)
    assert tr == [sample] and len(te) == 2 and fs == 123


def test_synthetic_amazon_schema():
    tr, te, fs = datasets.make_synthetic_amazon(n_samples=40, n_item=60, n_cate=6, n_user=30, max_len=16, seed=2)
    assert len(tr) == 28 and len(te) == 12 and fs == 96
    for label, useq, ulen, iseq, ilen in tr:
        assert label in (0, 1) and len(useq) == 16 and all(len(r) == 3 for r in useq)
        assert all(r == [0, 0, 0] for r in useq[:16 - ulen])              # front padding
        assert all(r[1] != 0 for r in useq[16 - ulen:])                   # real rows
        uid = useq[-1][0]
        assert all(r[0] == uid for r in useq[16 - ulen:])                 # uid constant per sample
        assert 66 <= uid < 96 and 60 <= useq[-1][2] < 66                  # id space: items, cates, users
    arr_tr, arr_te, fs2 = datasets.make_synthetic_amazon(n_samples=40, n_item=60, n_cate=6, n_user=30,
                                                         max_len=16, seed=2, as_arrays=True)
    np.testing.assert_array_equal(arr_tr["ids"], np.asarray([s[1] for s in tr]))
    assert fs2 == fs


def test_xlong_tsv_line_format_and_loader(tmp_path):
    path = str(tmp_path / "xl.txt")
    datasets.write_xlong_tsv(path, n_lines=5, seed=3, hist_len=1000)
    line = open(path).readline()
    f = line.rstrip("\n").split("\t")
    assert len(f) == 7 and len(f[2].split(",")) == 1000 and len(f[5].split(",")) == 184
    pos, neg, up, un = parse_xlong_line(line)
    assert pos.shape == (1001, 2) and neg.shape == (1001, 2)
    assert np.all(pos[:, 0] == int(f[1]) + 3269017)                      # uid + item_cnt
    np.testing.assert_array_equal(pos[:1000], neg[:1000])
    assert pos[1000, 1] == int(f[3]) and neg[1000, 1] == int(f[4])
    batches = list(DataLoader_Mul(path, 4))                               # 2 lines -> 4 rows per batch
    assert [len(b[1][0]) for b in batches] == [4, 4, 2]
    _, (label, ipart, ilen, upart, ulen) = batches[0]
    assert label == [1, 0, 1, 0] and ipart.shape == (4, 1001, 2) and upart.shape == (4, 184, 1)
    assert ilen == [1001] * 4 and ulen == [184] * 4
    ids, lab = datasets.make_synthetic_xlong_arrays(5, seed=3)
    np.testing.assert_array_equal(np.concatenate([b[1][1] for b in batches]), ids)
    assert lab.tolist() == [1, 0] * 5


def test_xlong_loader_reproduces_the_executed_reference_worker(tmp_path):
    """tests/golden/xlong_worker_reference.npz: synthetic TSV lines and what the reference's OWN
    ``DataLoader_Mul.worker`` (code/data_loader.py:47-89) returned for them, executed in the build container with Python-2
    ``map`` semantics restored (tests/golden/make_golden.py:make_xlong_fixture).  Our reader -- the streaming class and the
    whole-file staging with its array cache -- has to yield the same batches."""
    import os
    from hpmn_amd.data_loader import load_xlong_tsv
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xlong_worker_reference.npz"))
    path = str(tmp_path / "ref_lines.txt")
    with open(path, "w") as f:
        f.write("".join(z["lines"].tolist()))
    got = list(DataLoader_Mul(path, int(z["batchsize"])))
    assert len(got) == int(z["n_batches"]) == 3
    for b, (i, (label, ipart, ilen, upart, ulen)) in enumerate(got):
        assert i is None
        assert list(label) == z["b%d_label" % b].tolist()
        np.testing.assert_array_equal(ipart, z["b%d_item_part" % b])
        np.testing.assert_array_equal(upart, z["b%d_user_part" % b])
        assert list(ilen) == z["b%d_item_part_len" % b].tolist() and list(ulen) == z["b%d_user_part_len" % b].tolist()
    want_ids = np.concatenate([z["b%d_item_part" % b] for b in range(3)])
    want_user = np.concatenate([z["b%d_user_part" % b] for b in range(3)])
    want_label = np.concatenate([z["b%d_label" % b] for b in range(3)])
    for attempt, workers in ((0, 2), (1, 0)):                    # second call: served by the cache
        a = load_xlong_tsv(path, workers=workers, chunk_lines=2)
        assert a["ids"].dtype == np.int32
        np.testing.assert_array_equal(a["ids"], want_ids)
        np.testing.assert_array_equal(a["item_ids"], want_user)
        np.testing.assert_array_equal(a["label"], want_label)
        assert os.path.exists(path + ".hpmn_cache.npz")
    with open(path, "a") as f:                                    # the file changed: the cache is not trusted
        f.write(z["lines"].tolist()[0])
    assert load_xlong_tsv(path)["ids"].shape[0] == want_ids.shape[0] + 2
