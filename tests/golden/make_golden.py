"""Generate the committed golden fixtures.  Run in the BUILD container only:

    python tests/golden/make_golden.py

Two kinds of fixture, both plain data (inputs + expected outputs):

1. ``input_surface.json`` -- produced by EXECUTING THE REFERENCE's own code, which is only
   possible for the TF-free pieces:
     * ``DataLoader`` imported from /root/reference/code/data_loader.py (runs under py3);
     * ``front_padding`` AST-extracted from /root/reference/code/util.py:152-159 (the module
       itself imports tensorflow and cannot be imported).
   Nothing from /root/reference is copied into the repo: only the inputs we fed and the
   outputs the reference code returned are stored.

2. ``oracle_c0.npz`` / ``oracle_industry.npz`` -- fixed-seed weights + one batch + the
   outputs of the float64 NumPy restatement (oracle/hpmn_oracle.py).  PARITY UNPINNED by the
   reference (TF1.4 cannot run here); these freeze the restatement so that the GPU box,
   which has neither /root/reference nor any need to trust a re-run, checks against bytes
   committed from this container.
"""
import ast
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/code"


def reference_front_padding():
    src = open(os.path.join(REF, "util.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "front_padding"][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {}
    exec(compile(mod, "util.py:front_padding", "exec"), ns)
    return ns["front_padding"]


def make_input_surface():
    sys.path.insert(0, REF)
    import data_loader as ref_dl          # the reference module, executed here only
    rng = np.random.default_rng(7)
    fp = reference_front_padding()
    # raw (back-padded) samples in dataset.pkl form: (label, user_seq, user_len, item_seq, item_len)
    user_max, user_dim, item_max, item_dim = 6, 3, 4, 2
    raw = []
    for n in range(7):
        ul = int(rng.integers(1, user_max + 1))
        il = int(rng.integers(1, item_max + 1))
        useq = rng.integers(1, 50, size=(ul, user_dim)).tolist() + [[0] * user_dim] * (user_max - ul)
        iseq = rng.integers(1, 50, size=(il, item_dim)).tolist() + [[0] * item_dim] * (item_max - il)
        raw.append([int(rng.integers(0, 2)), useq, ul, iseq, il])
    padded = [list(fp(s, user_max, user_dim, item_max, item_dim)) for s in raw]
    batches = []
    for bs in (1, 3, 7, 10):
        out = []
        for i, (label, ipart, ilen, upart, ulen) in ref_dl.DataLoader(padded, bs):
            out.append(dict(i=int(i), label=[int(v) for v in label], item_part=np.asarray(ipart).tolist(),
                            item_part_len=[int(v) for v in ilen], user_part=np.asarray(upart).tolist(),
                            user_part_len=[int(v) for v in ulen]))
        batches.append(dict(batch_size=bs, batches=out))
    fixture = dict(user_max=user_max, user_dim=user_dim, item_max=item_max, item_dim=item_dim,
                   raw=raw, front_padded=padded, loader=batches,
                   source="reference DataLoader (code/data_loader.py:267-301) and front_padding "
                          "(code/util.py:152-159) executed in the build container")
    with open(os.path.join(HERE, "input_surface.json"), "w") as f:
        json.dump(fixture, f)


def make_oracle_vectors():
    from oracle import hpmn_oracle as O
    # C0 (Amazon sample: code/hpmn.py:576-595) on a small vocabulary
    cfg = O.HpmnConfig(feature_size=300, user_dim=3, user_maxlen=100, hidden_size=32, embedding_size=16,
                       hop=3, user_layers=(2, 2, 5, 5, 1), user_num_layers=3, industry=False, memory_reg=1e-5)
    rng = np.random.default_rng(O_SEED)
    B = 6
    ids = rng.integers(1, cfg.feature_size, size=(B, cfg.user_maxlen, cfg.user_dim)).astype(np.int32)
    for b, L in enumerate([100, 5, 37, 64, 1, 80]):        # ragged front padding, incl. extremes
        ids[b, :cfg.user_maxlen - L] = 0
    label = rng.integers(0, 2, size=B).astype(np.int32)
    p = O.randomize_params(O.init_params(cfg, seed=3), seed=4)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}   # fp32-representable
    out = O.forward(cfg, p, ids, label)
    np.savez_compressed(os.path.join(HERE, "oracle_c0.npz"), ids=ids, label=label,
                        **{"param:" + k: v.astype(np.float32) for k, v in p.items()},
                        memory=out["memory"], logit=out["logit"], prediction=out["prediction"],
                        memory_loss=out["memory_loss"], user_weights=out["user_weights"],
                        cross_entropy=out["cross_entropy"])
    # Industry graph (code/hpmn.py:284-296) at a reduced length: 41 + 23 zero steps = 64
    cfg2 = O.HpmnConfig(feature_size=500, user_dim=2, user_maxlen=41, hidden_size=64, embedding_size=16,
                        hop=3, user_layers=(2,) * 10 + (1,), user_num_layers=4, industry=True, memory_reg=5e-5)
    ids2 = rng.integers(0, cfg2.feature_size, size=(5, cfg2.user_maxlen, cfg2.user_dim)).astype(np.int32)
    label2 = rng.integers(0, 2, size=5).astype(np.int32)
    p2 = O.randomize_params(O.init_params(cfg2, seed=5), seed=6)
    p2 = {k: v.astype(np.float32).astype(np.float64) for k, v in p2.items()}
    out2 = O.forward(cfg2, p2, ids2, label2)
    np.savez_compressed(os.path.join(HERE, "oracle_industry.npz"), ids=ids2, label=label2,
                        **{"param:" + k: v.astype(np.float32) for k, v in p2.items()},
                        memory=out2["memory"], logit=out2["logit"], prediction=out2["prediction"],
                        memory_loss=out2["memory_loss"], user_weights=out2["user_weights"],
                        cross_entropy=out2["cross_entropy"])


# -----------------------------------------------------------------------------------------------------------------
# 3. preprocess fixtures: the reference's OWN remap / gen_user_item_group / gen_dataset executed on small synthetic
#    logs.  The scripts cannot be imported (module-level ``from util import ...`` pulls in
#    tensorflow; ``pickle`` is used under a name that is never bound), so the functions are AST-extracted -- exactly
#    like front_padding above -- and run in a namespace that supplies what the module level would have: pandas,
#    ``random`` (seeded as preprocess_amazon.py:12 does), the MAX_LEN constants, and a ``pickle``/``open`` pair that
#    keeps the dumped objects in memory instead of on disk.  Only inputs and returned outputs are stored.
# -----------------------------------------------------------------------------------------------------------------
class _MemPickle:
    """Stands in for the module name ``pickle`` / ``pkl`` inside the extracted functions: dump() keeps the object,
    load() hands the kept objects back in order."""
    HIGHEST_PROTOCOL = 2

    def __init__(self):
        self.objs, self.pos = [], 0

    def dump(self, obj, f, protocol=None):
        self.objs.append(obj)

    def load(self, f):
        obj = self.objs[self.pos]
        self.pos += 1
        return obj


class _NullFile:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _extract(script, names, extra):
    import random as _random

    import pandas as pd
    src = open(os.path.join(REF, script)).read()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(f.name for f in fns) == sorted(names)
    ns = dict(pd=pd, random=_random, open=_NullFile, print=lambda *a, **k: None)
    ns.update(extra)
    exec(compile(ast.Module(body=fns, type_ignores=[]), script, "exec"), ns)
    return ns


def synthetic_amazon_events(seed=11, n_user=40, n_item=25, n_cate=6):
    """Raw review events with string keys (as the Amazon dumps have), time ties inside a user, one user longer than
    MAX_LEN, one item touched by many users (so an item side longer than MAX_LEN exists for a later target)."""
    rng = np.random.default_rng(seed)
    items = ["B%05d" % i for i in range(n_item)]
    cate_of = {a: "cat%d" % int(rng.integers(0, n_cate)) for a in items}
    ev = []
    for u in range(n_user):
        n = 130 if u == 3 else int(rng.integers(2, 9))
        ts = np.sort(rng.integers(1000, 4000, size=n))          # a narrow range: equal times do occur
        for t in ts:
            a = items[0] if rng.random() < 0.35 else items[int(rng.integers(0, n_item))]
            ev.append(("U%03d" % u, a, cate_of[a], int(t)))
    # 110 extra one-off users who all touched item 0 early: its item side outgrows MAX_LEN
    for u in range(n_user, n_user + 110):
        ev.append(("U%03d" % u, items[0], cate_of[items[0]], int(rng.integers(10, 900))))
        ev.append(("U%03d" % u, items[int(rng.integers(1, n_item))], cate_of[items[1]], int(rng.integers(900, 1000))))
    order = rng.permutation(len(ev))                           # file order is not sorted
    return [ev[i] for i in order]


def make_preprocess_amazon():
    import random as _random

    import pandas as pd
    ev = synthetic_amazon_events()
    # categories must be a function of the item (aggregator joins them from the meta file): re-derive to be safe
    cate = {}
    for u, a, c, t in ev:
        cate.setdefault(a, c)
    ev = [(u, a, cate[a], t) for u, a, c, t in ev]
    df = pd.DataFrame(dict(reviewerID=[e[0] for e in ev], asin=[e[1] for e in ev],
                           unixReviewTime=[e[3] for e in ev], category=[e[2] for e in ev]))
    mem = _MemPickle()
    ns = _extract("preprocess_amazon.py", ["remap", "gen_user_item_group", "gen_dataset"],
                  dict(pickle=mem, pkl=mem, MAX_LEN=100, SAVE_PKL_PATH="save.pkl"))
    _random.seed(1111)                                          # preprocess_amazon.py:12
    df2, item_cnt, feature_size = ns["remap"](df)
    remapped = dict(uid=df2["reviewerID"].tolist(), iid=df2["asin"].tolist(), cid=df2["category"].tolist())
    ns["gen_user_item_group"](df2, item_cnt, feature_size)      # dumps user_df, item_df, item_cnt, feature_size
    ns["gen_dataset"]("save.pkl", "dataset.pkl")                # loads those four, dumps train, test, feature_size
    train, test, fs = mem.objs[4], mem.objs[5], mem.objs[6]
    fp = reference_front_padding()
    train_fp = [fp(s, 100, 3, 100, 2) for s in train]           # preprocess_amazon.py:331,335
    test_fp = [fp(s, 100, 3, 100, 2) for s in test]
    return dict(events=ev, remapped=remapped, item_cnt=int(item_cnt), feature_size=int(fs),
                train=_pack(train_fp), test=_pack(test_fp), seed=1111)


def synthetic_taobao_events(seed=12, n_user=30, n_item=40, n_cate=7):
    rng = np.random.default_rng(seed)
    tags = ["buy", "cart", "fav", "pv"]
    cate_of = rng.integers(100, 100 + n_cate, size=n_item)
    ev = []
    for u in range(n_user):
        n = 320 if u == 2 else int(rng.integers(2, 12))
        ts = np.sort(rng.integers(5000, 9000, size=n))
        for t in ts:
            i = 0 if rng.random() < 0.3 else int(rng.integers(0, n_item))
            ev.append((1000 + 7 * u, 50000 + 3 * i, int(cate_of[i]), tags[int(rng.integers(0, 4))], int(t)))
    for u in range(n_user, n_user + 45):                       # item 0's item side outgrows MAX_LEN_USER = 35
        ev.append((1000 + 7 * u, 50000, int(cate_of[0]), "pv", int(rng.integers(10, 4000))))
        ev.append((1000 + 7 * u, 50000 + 3 * int(rng.integers(1, n_item)), int(cate_of[1]), "pv", int(rng.integers(4000, 5000))))
    order = rng.permutation(len(ev))
    return [ev[i] for i in order]


def make_preprocess_taobao():
    import random as _random

    import pandas as pd
    ev = synthetic_taobao_events()
    df = pd.DataFrame(dict(uid=[e[0] for e in ev], iid=[e[1] for e in ev], cid=[e[2] for e in ev],
                           btag=[e[3] for e in ev], time=[e[4] for e in ev]))
    mem = _MemPickle()
    ns = _extract("preprocess_taobao.py", ["remap", "gen_user_item_group", "gen_dataset"],
                  dict(pkl=mem, pickle=mem, MAX_LEN_ITEM=300, MAX_LEN_USER=35))
    _random.seed(1111)                    # (preprocess_taobao.py never seeds: the fixture fixes the stream it ran with)
    df2, item_cnt, feature_size = ns["remap"](df)
    remapped = dict(uid=df2["uid"].tolist(), iid=df2["iid"].tolist(), cid=df2["cid"].tolist(), btag=df2["btag"].tolist())
    user_df, item_df = ns["gen_user_item_group"](df2, item_cnt, feature_size)
    ns["gen_dataset"](user_df, item_df, item_cnt, feature_size, "dataset.pkl")
    train, test, fs = mem.objs[0], mem.objs[1], mem.objs[2]
    fp = reference_front_padding()
    train_fp = [fp(s, 300, 4, 36, 3) for s in train]            # preprocess_taobao.py:211,215
    test_fp = [fp(s, 300, 4, 36, 3) for s in test]
    return dict(events=ev, remapped=remapped, item_cnt=int(item_cnt), feature_size=int(fs),
                train=_pack(train_fp), test=_pack(test_fp), seed=1111)


def _pack(samples):
    return dict(label=[int(s[0]) for s in samples],
                user=np.asarray([s[1] for s in samples], dtype=np.int64),
                user_len=[int(s[2]) for s in samples],
                item=np.asarray([s[3] for s in samples], dtype=np.int64),
                item_len=[int(s[4]) for s in samples])


def make_preprocess_fixtures():
    out = {}
    for name, fx in (("amazon", make_preprocess_amazon()), ("taobao", make_preprocess_taobao())):
        ev = fx["events"]
        for c in range(len(ev[0])):
            col = [e[c] for e in ev]
            out["%s_ev%d" % (name, c)] = np.asarray(col)
        for k, v in fx["remapped"].items():
            out["%s_remap_%s" % (name, k)] = np.asarray(v, dtype=np.int64)
        out[name + "_item_cnt"] = np.int64(fx["item_cnt"])
        out[name + "_feature_size"] = np.int64(fx["feature_size"])
        for split in ("train", "test"):
            for k, v in fx[split].items():
                out["%s_%s_%s" % (name, split, k)] = np.asarray(v, dtype=np.int64 if k != "user" and k != "item" else np.int32)
    np.savez_compressed(os.path.join(HERE, "preprocess_reference.npz"), **out)
    print("preprocess fixture: amazon %d+%d samples, taobao %d+%d samples" % (
        len(out["amazon_train_label"]), len(out["amazon_test_label"]),
        len(out["taobao_train_label"]), len(out["taobao_test_label"])))


# -----------------------------------------------------------------------------------------------------------------
# 4. XLong TSV lines through the reference's OWN DataLoader_Mul.worker (code/data_loader.py:47-89), in process: the
#    method is called unbound on a stand-in ``self`` whose queues are plain lists (the 1+8 process plumbing around it
#    moves lines, it does not change them).  Python-2 semantics are restored without touching the source: the module
#    global ``map`` is bound to a list-returning map (py2's), and the chunks are ``batchsize // 2`` lines (py2's
#    integer ``/``, data_loader.py:13).
# -----------------------------------------------------------------------------------------------------------------
class _Val:
    def __init__(self, v):
        self.value = v

    def get_lock(self):
        return _NullFile()


class _ListQueue:
    def __init__(self, items=()):
        self.items = list(items)

    def get(self, timeout=None):
        if not self.items:
            raise RuntimeError("empty")
        return self.items.pop(0)

    def put(self, x):
        self.items.append(x)

    def qsize(self):
        return 0


def synthetic_xlong_lines(n_lines=5, seed=21):
    rng = np.random.default_rng(seed)
    lines = []
    for k in range(n_lines):
        hist = rng.integers(1, 3269017, size=1000)
        up, un = rng.integers(0, 20000, size=184), rng.integers(0, 20000, size=184)
        lines.append("\t".join([str(100 + k), str(int(rng.integers(0, 20000))), ",".join(map(str, hist)),
                                str(int(rng.integers(1, 3269017))), str(int(rng.integers(1, 3269017))),
                                ",".join(map(str, up)), ",".join(map(str, un))]) + "\n")
    return lines


def make_xlong_fixture():
    import builtins
    sys.path.insert(0, REF)
    import data_loader as ref_dl
    ref_dl.map = lambda f, *it: list(builtins.map(f, *it))      # Python 2's map (data_loader.py:65,72-73 rely on it)
    lines = synthetic_xlong_lines()
    batchsize = 4
    per = batchsize // 2                                        # data_loader.py:13 under Python 2
    chunks = [lines[i:i + per] for i in range(0, len(lines), per)]

    class FakeSelf:
        pass
    me = FakeSelf()
    me.wait_time, me.max_q_size = 0.0, 10
    me.read_stop, me.work_qsize = _Val(1.0), _Val(len(chunks))
    me.qsize, me.work_stop = _Val(0.0), _Val(0.0)
    me.work, me.results = _ListQueue(chunks), _ListQueue()
    ref_dl.DataLoader_Mul.worker(me, 0)
    assert len(me.results.items) == len(chunks) and me.work_stop.value == 1.0
    out = dict(lines=np.asarray(lines), batchsize=np.int64(batchsize))
    for b, (i, data) in enumerate(me.results.items):
        label, item_part, item_part_len, user_part, user_part_len = data
        assert i is None
        out["b%d_label" % b] = np.asarray(label, dtype=np.int64)
        out["b%d_item_part" % b] = np.asarray(item_part, dtype=np.int64)
        out["b%d_item_part_len" % b] = np.asarray(item_part_len, dtype=np.int64)
        out["b%d_user_part" % b] = np.asarray(user_part, dtype=np.int64)
        out["b%d_user_part_len" % b] = np.asarray(user_part_len, dtype=np.int64)
    out["n_batches"] = np.int64(len(chunks))
    np.savez_compressed(os.path.join(HERE, "xlong_worker_reference.npz"), **out)
    print("xlong fixture: %d lines -> %d batches, item_part %s user_part %s" % (
        len(lines), len(chunks), out["b0_item_part"].shape, out["b0_user_part"].shape))


O_SEED = 20190521

if __name__ == "__main__":
    which = sys.argv[1:] or ["surface", "oracle", "preprocess", "xlong"]
    if "surface" in which:
        make_input_surface()
    if "oracle" in which:
        make_oracle_vectors()
    if "preprocess" in which:
        make_preprocess_fixtures()
    if "xlong" in which:
        make_xlong_fixture()
    print("wrote", sorted(os.listdir(HERE)))
