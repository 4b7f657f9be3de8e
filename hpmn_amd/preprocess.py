"""Raw interaction logs -> ``dataset_hpmn.pkl`` (the input format either side of the hot path).

What the reference's ``code/preprocess_amazon.py`` / ``code/preprocess_taobao.py`` produce, rebuilt as
one array-based pipeline (the reference scripts are Python-2 pandas loops that do not run as written --
``pickle`` is never imported under that name, py2-only file modes -- so they are the SPEC here, cited
line by line; their functions DO run when extracted, which is what pins this module, see below):

    events (uid, iid, extra columns, time)
      -> remap every id column into ONE shared id space, column after column, each column's values in
         sorted order (preprocess_amazon.py:51-67: items, categories, users;
         preprocess_taobao.py:26-48: items, users, categories, btags, +1 "unknown target btag")
      -> per user, events in time order; the LAST event is the prediction target
         (preprocess_amazon.py:137-150)
      -> train/test by the user's last-touch time against the 70th percentile of all users'
         (:127-134,:150)
      -> label: a coin flip keeps the real target (label 1) or replaces it with a random OTHER item and
         that item's category (label 0) (:151-158)
      -> user-side rows [uid, item, cate(, btag)] ending in the target row (:160-165), item-side rows
         [target_item, user(, btag)] of the users who touched the target item strictly earlier, ending in
         [target_item, uid(, btag)] (:167-181)
      -> keep the most recent ``max_len`` rows, FRONT-pad with all-zero rows
         (:189-197 + util.front_padding, code/util.py:152-159)
      -> sample = (label, user_rows [T][F], user_len, item_rows [T'][F'], item_len)

Pinned by executing the reference (r4): ``tests/golden/make_golden.py`` AST-extracts the reference's own ``remap`` /
``gen_user_item_group`` / ``gen_dataset`` of both scripts, runs them under Python 3 on small synthetic logs and commits
inputs + outputs; ``tests/test_preprocess.py`` requires this module to reproduce them exactly -- ids, rows, lengths,
labels, split and (Amazon) the shuffled order -- from the same ``random`` seed, which works because the draws are made
in the same order (one ``randint(0, 1)`` per user, ``randint(0, n_item - 1)`` until the item differs, two shuffles).
Python 2's ``random`` draws a different stream for the same seed, so a dataset built here is distributed like, not
identical to, one the reference built in 2019.

Besides the pickle (protocol 2, readable by the reference) ``write_dataset`` drops an ``.npz`` next to
it with the same samples as int32 arrays; ``load_dataset`` prefers it (the list-of-lists pickle of the
full Amazon set takes tens of seconds to parse and convert, the arrays map in instantly).
"""
from __future__ import annotations

import ast
import json
import os
import random
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import datasets


@dataclass
class Schema:
    """Which columns make up a row on either side and in which order the id spaces are laid out."""
    id_order: Tuple[str, ...]             # columns in id-space order; "iid" must come first (targets are drawn in [0, n_item))
    user_row: Tuple[str, ...]             # row of the user-side sequence, e.g. ("uid", "iid", "cid")
    item_row: Tuple[str, ...]             # row of the item-side sequence, e.g. ("iid", "uid")
    user_max: int
    item_max: int                         # rows kept when cropping the item side
    item_pad: int                         # rows after front padding (Taobao pads 35 kept rows to 36)
    unknown_btag: bool = False            # Taobao: one extra id is reserved for the target row's btag (feature_size += 1)
    # The reference writes that btag as ``feature_size`` itself (preprocess_taobao.py:131 after :48) -- ONE PAST the last row
    # of a [feature_size, E] table: TF's CPU gather raises on it, its GPU gather returns a zero row.  True reproduces the
    # reference's files byte for byte (pinned by tests/golden/preprocess_reference.npz, produced by executing the reference's
    # functions); False writes the reserved in-range id feature_size - 1 instead.
    unknown_btag_out_of_range: bool = True
    shuffle: bool = True                  # preprocess_amazon.py:205-206 shuffles, preprocess_taobao.py does not


AMAZON = Schema(id_order=("iid", "cid", "uid"), user_row=("uid", "iid", "cid"), item_row=("iid", "uid"),
                user_max=100, item_max=100, item_pad=100)
TAOBAO = Schema(id_order=("iid", "uid", "cid", "btag"), user_row=("uid", "iid", "cid", "btag"),
                item_row=("iid", "uid", "btag"), user_max=300, item_max=35, item_pad=36, unknown_btag=True,
                shuffle=False)


# ---------------------------------------------------------------------------------------
# raw file readers
# ---------------------------------------------------------------------------------------
def _records(path: str):
    """One dict per line: JSON, or the Python-literal dicts of the 2014 Amazon dumps
    (the reference ``eval``s each line, preprocess_amazon.py:24-32)."""
    with open(path, "r") as fin:
        for line in fin:
            line = line.strip()
            if not line:
                continue
            try:
                yield json.loads(line)
            except ValueError:
                yield ast.literal_eval(line)


def read_amazon(review_file: str, meta_file: str) -> Dict[str, np.ndarray]:
    """reviews (reviewerID, asin, unixReviewTime) joined with the item's category = last entry of its last
    category path, first meta record per asin wins (preprocess_amazon.py:34-49)."""
    cate = {}
    for r in _records(meta_file):
        cate.setdefault(r["asin"], r["categories"][-1][-1])
    uid, iid, cid, t = [], [], [], []
    for r in _records(review_file):
        uid.append(r["reviewerID"])
        iid.append(r["asin"])
        cid.append(cate[r["asin"]])
        t.append(int(r["unixReviewTime"]))
    return dict(uid=np.asarray(uid), iid=np.asarray(iid), cid=np.asarray(cid), time=np.asarray(t, dtype=np.int64))


def read_taobao(csv_file: str) -> Dict[str, np.ndarray]:
    """uid,iid,cid,btag,time rows without a header (preprocess_taobao.py:22-24)."""
    cols = dict(uid=[], iid=[], cid=[], btag=[], time=[])
    with open(csv_file, "r") as fin:
        for line in fin:
            parts = line.strip().split(",")
            if len(parts) != 5:
                continue
            cols["uid"].append(int(parts[0]))
            cols["iid"].append(int(parts[1]))
            cols["cid"].append(int(parts[2]))
            cols["btag"].append(parts[3])
            cols["time"].append(int(parts[4]))
    return {k: np.asarray(v) for k, v in cols.items()}


# ---------------------------------------------------------------------------------------
# the pipeline
# ---------------------------------------------------------------------------------------
def remap(events: Dict[str, np.ndarray], schema: Schema):
    """-> (int64 id columns, n_item, feature_size): one shared id space, column after column in
    ``schema.id_order``, every column's distinct values in sorted order."""
    out, base = {}, 0
    n_item = 0
    for col in schema.id_order:
        keys, inv = np.unique(events[col], return_inverse=True)
        out[col] = inv.astype(np.int64) + base
        if col == "iid":
            n_item = len(keys)
        base += len(keys)
    out["time"] = np.asarray(events["time"], dtype=np.int64)
    feature_size = base + (1 if schema.unknown_btag else 0)
    return out, n_item, feature_size


def _groups(key: np.ndarray, time: np.ndarray):
    """Stable (key, time) order and the [start, end) of every key's run, keys ascending -- the iteration
    order of ``df.sort_values([key, time]).groupby(key)``."""
    order = np.lexsort((time, key))
    ks = key[order]
    starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ends = np.r_[starts[1:], len(ks)]
    return order, ks[starts], starts, ends


def build_samples(ev: Dict[str, np.ndarray], n_item: int, feature_size: int, schema: Schema, seed: int = 1111):
    """-> (train, test) lists of front-padded samples."""
    rng = random.Random(seed)                                   # random.seed(1111), preprocess_amazon.py:12
    uorder, ukeys, ustart, uend = _groups(ev["uid"], ev["time"])
    iorder, ikeys, istart, iend = _groups(ev["iid"], ev["time"])
    item_slot = {int(k): n for n, k in enumerate(ikeys)}
    unknown = None
    if schema.unknown_btag:
        unknown = feature_size if schema.unknown_btag_out_of_range else feature_size - 1

    last_touch = ev["time"][uorder[uend - 1]]
    split_time = np.sort(last_touch)[int(len(last_touch) * 0.7)]

    extra_user = [c for c in schema.user_row if c not in ("uid", "iid")]          # cid (, btag)
    train, test = [], []
    for g in range(len(ukeys)):
        rows = uorder[ustart[g]:uend[g]]
        uid = int(ukeys[g])
        t_target = int(ev["time"][rows[-1]])
        target = int(ev["iid"][rows[-1]])
        target_extra = {c: int(ev[c][rows[-1]]) for c in extra_user}
        if "btag" in target_extra:
            target_extra["btag"] = unknown                      # preprocess_taobao.py:131
        label = 1
        if rng.randint(0, 1) == 1:                               # negative sample: a random OTHER item
            label = 0
            real = target
            while target == real:
                target = rng.randint(0, n_item - 1)
            first = iorder[istart[item_slot[target]]]
            target_extra["cid"] = int(ev["cid"][first])          # that item's category (:157)

        def user_row(r):
            return [uid if c == "uid" else int(ev[c][r]) for c in schema.user_row]

        urows = [user_row(r) for r in rows[:-1]]
        urows.append([uid if c == "uid" else target if c == "iid" else target_extra[c] for c in schema.user_row])

        # item side: who touched the target item strictly before the target time
        s = item_slot[target]
        irows = []
        for r in iorder[istart[s]:iend[s]]:
            if int(ev["time"][r]) < t_target:
                irows.append([target if c == "iid" else int(ev[c][r]) for c in schema.item_row])
        irows.append([target if c == "iid" else uid if c == "uid" else unknown for c in schema.item_row])

        urows = urows[-schema.user_max:]
        irows = irows[-schema.item_max:]
        sample = datasets.front_padding((label, urows, len(urows), irows, len(irows)), schema.user_max,
                                        len(schema.user_row), schema.item_pad, len(schema.item_row))
        (test if t_target > split_time else train).append(sample)
    if schema.shuffle:
        rng.shuffle(train)
        rng.shuffle(test)
    return train, test


def to_arrays(samples: Sequence) -> Dict[str, np.ndarray]:
    """list-of-samples -> the dict-of-arrays form ``Hpmn`` accepts directly (ids = user side)."""
    return dict(label=np.asarray([s[0] for s in samples], dtype=np.int32),
                ids=np.asarray([s[1] for s in samples], dtype=np.int32),
                length=np.asarray([s[2] for s in samples], dtype=np.int32),
                item_ids=np.asarray([s[3] for s in samples], dtype=np.int32),
                item_length=np.asarray([s[4] for s in samples], dtype=np.int32))


def write_dataset(path: str, train, test, feature_size: int):
    """``dataset_hpmn.pkl`` (three consecutive pickles, code/hpmn.py:571-575) + the array cache."""
    datasets.save_dataset_pkl(path, train, test, feature_size, protocol=2)
    tr, te = to_arrays(train), to_arrays(test)
    np.savez(os.path.splitext(path)[0] + ".npz", feature_size=np.int64(feature_size),
             **{"train_" + k: v for k, v in tr.items()}, **{"test_" + k: v for k, v in te.items()})


def load_dataset(path: str):
    """-> (trainset, testset, feature_size) for ``Hpmn``: the array cache when it is there and not older
    than the pickle, else the pickle itself (and the cache is written for next time)."""
    cache = os.path.splitext(path)[0] + ".npz"
    if os.path.exists(cache) and (not os.path.exists(path) or os.path.getmtime(cache) >= os.path.getmtime(path)):
        z = np.load(cache)
        pick = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
        return pick("train_"), pick("test_"), int(z["feature_size"])
    train, test, feature_size = datasets.load_dataset_pkl(path)
    try:
        np.savez(cache, feature_size=np.int64(feature_size),
                 **{"train_" + k: v for k, v in to_arrays(train).items()},
                 **{"test_" + k: v for k, v in to_arrays(test).items()})
    except OSError:
        pass                                                     # read-only data directory: just skip the cache
    return train, test, feature_size


def preprocess_amazon(review_file: str, meta_file: str, out_pkl: str, seed: int = 1111):
    ev, n_item, fs = remap(read_amazon(review_file, meta_file), AMAZON)
    train, test = build_samples(ev, n_item, fs, AMAZON, seed)
    write_dataset(out_pkl, train, test, fs)
    return len(train), len(test), fs


def preprocess_taobao(csv_file: str, out_pkl: str, seed: int = 1111, in_range_unknown_btag: bool = True):
    """``in_range_unknown_btag`` (default): the target rows' btag is written as feature_size - 1, the reserved id
    INSIDE the table, so that `preprocess taobao` -> `hpmn.py taobao` runs as it stands.  False reproduces the reference's files
    byte for byte: its target rows carry btag == feature_size (preprocess_taobao.py:48,131), one past the table -- TF's CPU
    gather raises on that id, its GPU gather returns a zero row; the loader then needs HPMN_OOB_IDS=zero
    (see Schema.unknown_btag_out_of_range; the byte-exact mode is what tests/golden/preprocess_reference.npz pins)."""
    import dataclasses
    schema = dataclasses.replace(TAOBAO, unknown_btag_out_of_range=not in_range_unknown_btag)
    ev, n_item, fs = remap(read_taobao(csv_file), schema)
    train, test = build_samples(ev, n_item, fs, schema, seed)
    write_dataset(out_pkl, train, test, fs)
    return len(train), len(test), fs


def main(argv: Optional[List[str]] = None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description="raw logs -> dataset_hpmn.pkl (+ .npz array cache)")
    sub = ap.add_subparsers(dest="which", required=True)
    a = sub.add_parser("amazon")
    a.add_argument("--reviews", default="../data/raw_data/amazon/Electronics_5.json")     # preprocess_amazon.py:14-15
    a.add_argument("--meta", default="../data/raw_data/amazon/meta_Electronics.json")
    a.add_argument("--out", default="../data/amazon/dataset_hpmn.pkl")
    t = sub.add_parser("taobao")
    t.add_argument("--csv", default="../data/raw_data/taobao/taobao_sample.csv")         # preprocess_taobao.py:12
    t.add_argument("--out", default="../data/taobao/dataset_hpmn.pkl")
    t.add_argument("--reference-unknown-btag", action="store_true",
                   help="target btag = feature_size, ONE PAST the table, exactly as the reference writes it (the loader then "
                        "needs HPMN_OOB_IDS=zero); default: the in-range id feature_size - 1")
    args = ap.parse_args(argv)
    if args.which == "amazon":
        print("train %d test %d feature_size %d" % preprocess_amazon(args.reviews, args.meta, args.out))
    else:
        print("train %d test %d feature_size %d" % preprocess_taobao(args.csv, args.out, in_range_unknown_btag=not args.reference_unknown_btag))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
