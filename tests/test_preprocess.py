"""Raw logs -> dataset_hpmn.pkl (hpmn_amd/preprocess.py).

PINNED BY REFERENCE EXECUTION (r4): tests/golden/preprocess_reference.npz holds synthetic raw events and what the
reference's own ``remap`` / ``gen_user_item_group`` / ``gen_dataset`` (code/preprocess_amazon.py:51-67,104-212;
code/preprocess_taobao.py:26-57,109-189) + ``front_padding`` (code/util.py:152-159) returned for them when executed in
the build container (tests/golden/make_golden.py); this module has to reproduce every id, row, length, label, the
train/test split and the shuffled order exactly.  Plus schema invariants on small synthetic raw files."""
import json
import os

import numpy as np

from hpmn_amd import datasets, preprocess as P


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_reference.npz")


def _check_split(z, pre, samples):
    assert [s[0] for s in samples] == z[pre + "label"].tolist()
    np.testing.assert_array_equal(np.asarray([s[1] for s in samples]), z[pre + "user"])
    np.testing.assert_array_equal(np.asarray([s[3] for s in samples]), z[pre + "item"])
    assert [s[2] for s in samples] == z[pre + "user_len"].tolist()
    assert [s[4] for s in samples] == z[pre + "item_len"].tolist()


def test_amazon_port_reproduces_the_executed_reference():
    """remap: same ids for every event; build_samples from random.seed(1111) (preprocess_amazon.py:12): the same samples in
    the same (shuffled) order as the reference's gen_dataset -- incl. equal times inside a user, a user and an item side
    longer than MAX_LEN, negative targets and their categories."""
    z = np.load(GOLD)
    ev = dict(uid=z["amazon_ev0"], iid=z["amazon_ev1"], cid=z["amazon_ev2"], time=z["amazon_ev3"])
    out, n_item, fs = P.remap(ev, P.AMAZON)
    assert n_item == int(z["amazon_item_cnt"]) and fs == int(z["amazon_feature_size"])
    for k in ("uid", "iid", "cid"):
        np.testing.assert_array_equal(out[k], z["amazon_remap_" + k])
    train, test = P.build_samples(out, n_item, fs, P.AMAZON, seed=1111)
    assert len(train) == len(z["amazon_train_label"]) > 50 and len(test) == len(z["amazon_test_label"]) > 20
    _check_split(z, "amazon_train_", train)
    _check_split(z, "amazon_test_", test)
    assert max(z["amazon_train_user_len"].max(), z["amazon_test_user_len"].max()) == 100      # the fixture does crop
    assert max(z["amazon_train_item_len"].max(), z["amazon_test_item_len"].max()) == 100
    assert 0 in z["amazon_train_label"] and 1 in z["amazon_train_label"]


def test_taobao_port_reproduces_the_executed_reference():
    """The Taobao pair: id order items, users, categories, btags, +1; no shuffle; 35 item-side rows padded to 36; and the
    reference's target btag = feature_size (one past the table -- reproduced as written; the in-range variant is a switch)."""
    import dataclasses
    z = np.load(GOLD)
    ev = dict(uid=z["taobao_ev0"], iid=z["taobao_ev1"], cid=z["taobao_ev2"], btag=z["taobao_ev3"], time=z["taobao_ev4"])
    out, n_item, fs = P.remap(ev, P.TAOBAO)
    assert n_item == int(z["taobao_item_cnt"]) and fs == int(z["taobao_feature_size"])
    for k in ("uid", "iid", "cid", "btag"):
        np.testing.assert_array_equal(out[k], z["taobao_remap_" + k])
    train, test = P.build_samples(out, n_item, fs, P.TAOBAO, seed=1111)
    _check_split(z, "taobao_train_", train)
    _check_split(z, "taobao_test_", test)
    assert int(z["taobao_train_user"].max()) == fs                            # the reference's out-of-range unknown btag
    assert max(z["taobao_train_user_len"].max(), z["taobao_test_user_len"].max()) == 300
    assert max(z["taobao_train_item_len"].max(), z["taobao_test_item_len"].max()) == 35
    # the switch: identical except that id
    fixed = dataclasses.replace(P.TAOBAO, unknown_btag_out_of_range=False)
    train2, _ = P.build_samples(out, n_item, fs, fixed, seed=1111)
    a, b = np.asarray([s[1] for s in train]), np.asarray([s[1] for s in train2])
    assert (a != b).sum() == len(train) and set(b[a != b].tolist()) == {fs - 1} and int(b.max()) < fs


def _write_amazon(tmp, n_user=40, n_item=25, n_cate=6, seed=3):
    rng = np.random.default_rng(seed)
    items = ["B%05d" % i for i in range(n_item)]
    cate_of = {a: "cat%d" % rng.integers(0, n_cate) for a in items}
    meta = os.path.join(tmp, "meta.json")
    with open(meta, "w") as f:
        for k, a in enumerate(items):
            rec = {"asin": a, "categories": [["Electronics", "x"], ["Electronics", cate_of[a]]]}
            f.write((json.dumps(rec) if k % 2 else repr(rec)) + "\n")       # JSON and python-literal lines
        f.write(json.dumps({"asin": items[0], "categories": [["dup", "ignored"]]}) + "\n")   # first record wins
    rev = os.path.join(tmp, "reviews.json")
    events = []
    with open(rev, "w") as f:
        for u in range(n_user):
            n = int(rng.integers(2, 9)) if u else 130                      # user 0 is longer than max_len
            ts = np.sort(rng.choice(np.arange(1000, 100000), size=n, replace=False))
            for t in ts:
                a = items[int(rng.integers(0, n_item))]
                events.append(("U%03d" % u, a, int(t)))
                f.write(json.dumps({"reviewerID": "U%03d" % u, "asin": a, "unixReviewTime": int(t)}) + "\n")
    return rev, meta, events, cate_of


def test_amazon_pipeline_invariants(tmp_path):
    rev, meta, events, cate_of = _write_amazon(str(tmp_path))
    out = str(tmp_path / "amazon" / "dataset_hpmn.pkl")
    ntr, nte, fs = P.preprocess_amazon(rev, meta, out)
    users = sorted({e[0] for e in events})
    items = sorted({e[1] for e in events})
    cates = sorted({cate_of[a] for a in items})
    assert ntr + nte == len(users)
    assert fs == len(items) + len(cates) + len(users)                       # one id space: items, categories, users
    item_id = {a: i for i, a in enumerate(items)}
    cate_id = {c: len(items) + i for i, c in enumerate(cates)}
    user_id = {u: len(items) + len(cates) + i for i, u in enumerate(users)}

    train, test, fs2 = datasets.load_dataset_pkl(out)
    assert fs2 == fs and len(train) == ntr and len(test) == nte
    by_user = {}
    for u, a, t in events:
        by_user.setdefault(u, []).append((t, a))
    last_touch = sorted(max(t for t, _ in v) for v in by_user.values())
    split = last_touch[int(len(last_touch) * 0.7)]
    seen = set()
    n_neg = 0
    for which, samples in (("train", train), ("test", test)):
        for label, urows, ulen, irows, ilen in samples:
            assert len(urows) == 100 and len(irows) == 100 and all(len(r) == 3 for r in urows) and all(len(r) == 2 for r in irows)
            assert urows[:100 - ulen] == [[0, 0, 0]] * (100 - ulen) and irows[:100 - ilen] == [[0, 0]] * (100 - ilen)
            uid = urows[-1][0]
            uname = users[uid - len(items) - len(cates)]
            seen.add(uname)
            hist = sorted(by_user[uname])
            assert ulen == min(len(hist), 100)
            assert (max(t for t, _ in hist) > split) == (which == "test")
            want_hist = [[uid, item_id[a], cate_id[cate_of[a]]] for _, a in hist[:-1]][-(ulen - 1):] if ulen > 1 else []
            assert urows[100 - ulen:-1] == want_hist
            target, tcate = urows[-1][1], urows[-1][2]
            assert tcate == cate_id[cate_of[items[target]]]
            if label == 1:
                assert target == item_id[hist[-1][1]]
            else:
                n_neg += 1
                assert target != item_id[hist[-1][1]]
            t_target = hist[-1][0]
            # item side: users who touched the target strictly earlier, in time order, then this user
            earlier = sorted((t, user_id[u]) for u, a, t in events if item_id[a] == target and t < t_target)
            want_item = ([[target, uu] for _, uu in earlier] + [[target, uid]])[-100:]
            assert irows[100 - ilen:] == want_item and ilen == len(want_item)
    assert seen == set(users) and 0 < n_neg < len(users)

    # array cache: same samples, accepted as-is by the model's dataset wrapper
    tr, te, fs3 = P.load_dataset(out)
    assert fs3 == fs and tr["ids"].dtype == np.int32 and tr["ids"].shape == (ntr, 100, 3)
    np.testing.assert_array_equal(tr["ids"], np.asarray([s[1] for s in train]))
    np.testing.assert_array_equal(te["label"], np.asarray([s[0] for s in test]))
    np.testing.assert_array_equal(tr["length"], np.asarray([s[2] for s in train]))
    os.remove(os.path.splitext(out)[0] + ".npz")                             # cache gone: rebuilt from the pickle
    tr2, _, _ = P.load_dataset(out)
    np.testing.assert_array_equal(np.asarray([s[1] for s in tr2]), tr["ids"])
    assert os.path.exists(os.path.splitext(out)[0] + ".npz")


def test_taobao_pipeline_invariants(tmp_path):
    rng = np.random.default_rng(5)
    path = str(tmp_path / "taobao.csv")
    rows = []
    with open(path, "w") as f:
        for u in range(30):
            n = int(rng.integers(2, 12)) if u else 320                      # user 0 exceeds the 300-row window
            ts = np.sort(rng.choice(np.arange(10, 10 ** 6), size=n, replace=False))
            for t in ts:
                iid = int(rng.integers(1000, 1012)) if u else int(rng.integers(1000, 1002))
                r = (100 + u, iid, 7 + iid % 3, ["pv", "buy", "cart", "fav"][int(rng.integers(0, 4))], int(t))
                rows.append(r)
                f.write("%d,%d,%d,%s,%d\n" % r)
    out = str(tmp_path / "taobao" / "dataset_hpmn.pkl")
    ntr, nte, fs = P.preprocess_taobao(path, out)       # (the default writes the in-range unknown btag: ADVICE r4)
    exact = str(tmp_path / "taobao_exact" / "dataset_hpmn.pkl")
    assert P.main(["taobao", "--csv", path, "--out", exact, "--reference-unknown-btag"]) == 0
    tr_exact, te_exact, fs_exact = datasets.load_dataset_pkl(exact)
    assert fs_exact == fs and max(max(r[3] for r in s[1]) for s in tr_exact + te_exact) == fs        # one past the table
    n_item, n_user = len({r[1] for r in rows}), len({r[0] for r in rows})
    n_cate, n_btag = len({r[2] for r in rows}), len({r[3] for r in rows})
    assert fs == n_item + n_user + n_cate + n_btag + 1 and ntr + nte == n_user   # items, users, categories, btags, +1
    train, test, _ = datasets.load_dataset_pkl(out)
    for label, urows, ulen, irows, ilen in train + test:
        assert len(urows) == 300 and len(irows) == 36 and len(urows[0]) == 4 and len(irows[0]) == 3
        assert ilen <= 35 and irows[0] == [0, 0, 0]                          # 35 rows kept, padded to 36
        assert urows[-1][3] == fs - 1 and irows[-1][2] == fs - 1             # unknown btag of the target row
        assert 0 <= urows[-1][1] < n_item and n_item <= urows[-1][0] < n_item + n_user
        assert n_item + n_user <= urows[-1][2] < n_item + n_user + n_cate
        assert all(r == [0, 0, 0, 0] for r in urows[:300 - ulen]) and all(r[0] == urows[-1][0] for r in urows[300 - ulen:])
        assert irows[-1][:2] == [urows[-1][1], urows[-1][0]]
    assert max(s[2] for s in train + test) == 300                            # the long user was cropped to the window
