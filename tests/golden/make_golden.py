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


O_SEED = 20190521

if __name__ == "__main__":
    make_input_surface()
    make_oracle_vectors()
    print("wrote", sorted(os.listdir(HERE)))
