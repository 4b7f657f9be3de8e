"""r6 debugging: two gloo ranks on one GPU, evaluation by dataset shard -- where do the gathered predictions go wrong?"""
import os, sys, tempfile
import numpy as np, torch, torch.distributed as td
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dp_worker
from hpmn_amd import dist
td.init_process_group("gloo"); torch.cuda.set_device(0)
rank, world = td.get_rank(), td.get_world_size()
m, te, _ = dp_worker.build_eval(tempfile.mkdtemp())
ds = m._dev(te)
a, b = dist.shard_bounds(0, ds.n, rank, world)
tile = m.forward_inference(ds.ids[a:b], want_logit=False, want_att=False)["prediction"]
m.TILED_EVAL_MIN_ROWS = 0
ref_all = m.forward_inference(ds.ids)["prediction"]
m.TILED_EVAL_MIN_ROWS = 1536
torch.cuda.synchronize()
print("rank", rank, "shard", (a, b), "tile path vs per-sequence kernels on my shard: max diff %.3g" % float((tile - ref_all[a:b]).abs().max()), flush=True)
g = dist.gather_predictions(tile.contiguous(), ds.n)
torch.cuda.synchronize()
print("rank", rank, "gathered vs reference: max diff %.3g; first half %.3g second half %.3g" % (
    float((g - ref_all).abs().max()), float((g[:ds.n // 2] - ref_all[:ds.n // 2]).abs().max()), float((g[ds.n // 2:] - ref_all[ds.n // 2:]).abs().max())), flush=True)
print("rank", rank, "eval", m.eval(te, 500), "param checksum %.6f" % float(m.flat_param[m.params["Embedding/emb_mtx"].numel():].double().sum()), flush=True)
td.barrier(); td.destroy_process_group()
