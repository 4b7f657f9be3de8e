"""dataset_hpmn.pkl I/O and synthetic datasets with the reference schema.

File format (reference: /root/reference/code/preprocess_amazon.py:319-341, reader
code/hpmn.py:571-575): three consecutive pickles -- train list, test list, feature_size.
Each sample is ``(label, user_seq, user_len, item_seq, item_len)`` where ``user_seq`` is a
list of ``user_maxlen`` rows of ``user_dim`` ints, FRONT-padded with all-zero rows
(code/util.py:152-159); the last row is the target.  Amazon rows are
``[uid, item_id, cate_id]``; id space = items, then categories, then users
(code/preprocess_amazon.py:51-67).  The reference writes protocol-0 Python-2 pickles;
``load_dataset_pkl`` reads those (latin1) as well as our own.

The real data files are absent, so every benchmark/test dataset is
synthetic: seeds ``numpy.random.default_rng(20190521 + cfg)``.
"""
from __future__ import annotations

import os
import pickle
from typing import List, Tuple

import numpy as np

SEED_BASE = 20190521


def front_padding(seq, user_max, user_dim, item_max, item_dim):
    """Back-padded sample -> front-padded sample (code/util.py:152-159)."""
    label, user_seq, user_len, item_seq, item_len = seq
    user_seq = [[0] * user_dim for _ in range(user_max - user_len)] + list(user_seq[:user_len])
    item_seq = [[0] * item_dim for _ in range(item_max - item_len)] + list(item_seq[:item_len])
    return label, user_seq, user_len, item_seq, item_len


def load_dataset_pkl(path: str):
    """-> (trainset, testset, feature_size); code/hpmn.py:571-575.  Python-2 protocol-0/2
    pickles (what the reference's preprocess scripts write) are read with latin1 strings."""
    with open(path, "rb") as fin:
        return tuple(pickle.load(fin, encoding="latin1") for _ in range(3))


def save_dataset_pkl(path: str, trainset, testset, feature_size, protocol: int = 2):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as fout:
        pickle.dump(trainset, fout, protocol=protocol)
        pickle.dump(testset, fout, protocol=protocol)
        pickle.dump(feature_size, fout, protocol=protocol)


# ---------------------------------------------------------------------------------------
# synthetic Amazon/Taobao-style datasets (in-memory list format of code/data_loader.py:267)
# ---------------------------------------------------------------------------------------
def make_synthetic_amazon(n_samples=3000, n_item=2000, n_cate=50, n_user=3000, max_len=100,
                          user_dim=3, item_max=100, seed=SEED_BASE, train_frac=0.7, as_arrays=False):
    """Synthetic ``dataset_hpmn.pkl`` content (C0/C1 of SURVEY.md section 8d).

    Lengths ~ 5 + Geometric(0.25) capped at max_len (5-core => >= 5 events); items Zipf(1.1);
    category = fixed map item -> cate; uid constant per sample; label Bernoulli(.5) with a
    planted signal: the positive target is drawn from the user's modal category, the negative
    from a different one (history-dependent part); favourites concentrate on the lower half of the
    categories and negatives on the upper half (marginal part, learnable from the target row
    alone within ~100 steps), so AUC is learnable.  user_dim 3 rows = [uid, item, cate]; user_dim 4
    (Taobao) adds a behaviour tag.  Returns (trainset, testset, feature_size) or, with
    ``as_arrays``, dict(ids [N,T,F] int32, label [N] int32, length [N]) per split.
    """
    rng = np.random.default_rng(seed)
    item_cate = rng.integers(0, n_cate, size=n_item)
    cate_items = [np.nonzero(item_cate == c)[0] for c in range(n_cate)]
    cate_items = [ci if len(ci) else np.array([0]) for ci in cate_items]
    zipf_p = 1.0 / np.arange(1, n_item + 1) ** 1.1
    zipf_p /= zipf_p.sum()
    n_btag = 5 if user_dim == 4 else 0
    feature_size = n_item + n_cate + n_user + n_btag
    off_c, off_u, off_b = n_item, n_item + n_cate, n_item + n_cate + n_user

    ids = np.zeros((n_samples, max_len, user_dim), dtype=np.int32)
    labels = rng.integers(0, 2, size=n_samples).astype(np.int32)
    lengths = np.minimum(5 + rng.geometric(0.25, size=n_samples), max_len).astype(np.int32)
    for n in range(n_samples):
        L = int(lengths[n])
        uid = off_u + int(rng.integers(0, n_user))
        # marginal signal: users mostly favour the "attractive" lower half of the categories ...
        half = max(1, n_cate // 2)
        fav = int(rng.integers(0, half)) if rng.random() < 0.85 else int(rng.integers(0, n_cate))
        hist = rng.choice(n_item, size=L - 1, p=zipf_p)
        # half of the history from the favourite category -> modal category is recoverable
        k = (L - 1 + 1) // 2
        hist[:k] = rng.choice(cate_items[fav], size=k)
        rng.shuffle(hist)
        if labels[n] == 1:
            tgt = int(rng.choice(cate_items[fav]))
        else:
            # ... and negatives mostly come from the upper half (never from the favourite category)
            other = int(rng.integers(half, n_cate)) if rng.random() < 0.85 else int(rng.integers(0, n_cate))
            if other == fav:
                other = (fav + 1) % n_cate
            tgt = int(rng.choice(cate_items[other]))
        items = np.concatenate([hist, [tgt]]).astype(np.int64)
        # item id 0 is the padding id: shift real items into [1, n_item)
        items = np.maximum(items, 1)
        row = ids[n, max_len - L:]
        row[:, 0] = uid
        row[:, 1] = items
        row[:, 2] = off_c + item_cate[items]
        if user_dim == 4:
            row[:, 3] = off_b + rng.integers(0, n_btag, size=L)
    n_train = int(n_samples * train_frac)
    if as_arrays:
        mk = lambda sl: dict(ids=ids[sl], label=labels[sl], length=lengths[sl])
        return mk(slice(0, n_train)), mk(slice(n_train, None)), feature_size

    def to_list(sl):
        out = []
        for n in range(*sl.indices(n_samples)):
            L = int(lengths[n])
            item_seq = [[0, 0]] * (item_max - 1) + [[int(ids[n, -1, 1]), int(ids[n, -1, 0])]]
            out.append((int(labels[n]), ids[n].tolist(), L, item_seq, 1))
        return out

    return to_list(slice(0, n_train)), to_list(slice(n_train, n_samples)), feature_size


# ---------------------------------------------------------------------------------------
# synthetic XLong (DataLoader_Mul TSV format, code/data_loader.py:57-85)
# ---------------------------------------------------------------------------------------
XLONG_ITEM_CNT = 3269017
XLONG_PV_CNT = 19002
XLONG_USERS = 20000


def xlong_feature_size(item_cnt=XLONG_ITEM_CNT, users=XLONG_USERS, pv=XLONG_PV_CNT):
    # code/hpmn.py:630-632: pv_cnt + rows(graph_emb) + 20000
    return pv + item_cnt + users


def make_synthetic_xlong_arrays(n_lines, seed=SEED_BASE + 3, item_cnt=XLONG_ITEM_CNT, users=XLONG_USERS,
                                hist_len=1000, n_cluster=64):
    """Synthetic XLong rows already in loader output form: ids [2*n_lines, hist_len+1, 2] int32,
    label [2*n_lines].  Row layout [uid + item_cnt, item] (code/data_loader.py:66-70); each
    line yields a positive and a negative row sharing the history (:75-80).  Planted signal:
    items are grouped in ``n_cluster`` id-blocks; a user's history concentrates on two blocks and
    the positive target comes from them."""
    rng = np.random.default_rng(seed)
    ids = np.empty((2 * n_lines, hist_len + 1, 2), dtype=np.int32)
    label = np.tile(np.array([1, 0], dtype=np.int32), n_lines)
    block = item_cnt // n_cluster
    for n in range(n_lines):
        uid = int(rng.integers(0, users)) + item_cnt
        fav = rng.integers(0, n_cluster, size=2)
        which = rng.integers(0, 4, size=hist_len)
        base = np.where(which < 3, fav[which % 2], rng.integers(0, n_cluster, size=hist_len))
        hist = base * block + rng.integers(0, block, size=hist_len)
        pos = int(fav[0]) * block + int(rng.integers(0, block))
        other = (int(fav[0]) + 1 + int(rng.integers(0, n_cluster - 1))) % n_cluster
        if other == fav[1]:
            other = (other + 1) % n_cluster
        neg = other * block + int(rng.integers(0, block))
        for k, tgt in enumerate((pos, neg)):
            r = ids[2 * n + k]
            r[:, 0] = uid
            r[:hist_len, 1] = hist
            r[hist_len, 1] = tgt
    return ids, label


def make_synthetic_graph_emb(seed=SEED_BASE + 4, item_cnt=XLONG_ITEM_CNT, n_cluster=64, dim=16, spread=0.05):
    """Stand-in for ``graph_emb.npy`` (code/hpmn.py:631-635: pre-trained item embeddings that
    initialise the table): one centroid per id-block of ``make_synthetic_xlong_arrays`` plus
    N(0, spread) noise, so the planted "target comes from the user's favourite block" signal is
    visible to the model the way graph-trained embeddings make item similarity visible."""
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((n_cluster, dim)).astype(np.float32) * 0.3
    block = item_cnt // n_cluster
    cl = np.minimum(np.arange(item_cnt) // block, n_cluster - 1)
    emb = cent[cl]
    emb += rng.standard_normal((item_cnt, dim), dtype=np.float32) * spread
    return emb


def write_xlong_tsv(path, n_lines, seed=SEED_BASE + 3, item_cnt=XLONG_ITEM_CNT, users=XLONG_USERS,
                    hist_len=1000, user_part_len=184):
    """Write ``n_lines`` lines in the reference's TSV format (code/data_loader.py:59-73)."""
    ids, _ = make_synthetic_xlong_arrays(n_lines, seed, item_cnt, users, hist_len)
    rng = np.random.default_rng(seed + 1)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        for n in range(n_lines):
            pos, neg = ids[2 * n], ids[2 * n + 1]
            uid = int(pos[0, 0]) - item_cnt
            hist = ",".join(str(int(v)) for v in pos[:hist_len, 1])
            up = ",".join(str(int(v)) for v in rng.integers(0, users, size=user_part_len))
            un = ",".join(str(int(v)) for v in rng.integers(0, users, size=user_part_len))
            f.write("%d\t%d\t%s\t%d\t%d\t%s\t%s\n" % (n, uid, hist, int(pos[hist_len, 1]),
                                                     int(neg[hist_len, 1]), up, un))
