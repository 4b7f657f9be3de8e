"""Evaluation throughput at H = 128 (C4 shape): the tile kernel (hpmn_tile_fwd) against the per-sequence scans."""
import os, sys, time, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hpmn_amd import ops
dev = torch.device("cuda:0")
c = dict(bench.CONFIGS["c4"]); c["config_id"] = "c4"
m = bench.build_model(c, tempfile.mkdtemp(), dev, seed=0)
for rows in (2000, 4096):
    ids = torch.cat([b[0] for b in bench.synth_batches(c, (rows + 499) // 500, 500, 7, dev)], 0)[:rows]
    for name, fn in (("tile128", lambda: ops.tiled_forward_inference(m.spec, ids, m.params["Embedding/emb_mtx"], m._gru_weights())),
                     ("per-sequence", lambda: ops.scan_forward_inference(m.spec, ids, m.params["Embedding/emb_mtx"], m._gru_weights()))):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("%5d rows  %-13s %.2f ms  %.0f k sequences/s" % (rows, name, dt * 1e3, rows / dt / 1e3), flush=True)
