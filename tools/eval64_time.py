"""Evaluation throughput at H = 64 (C3 / C2 shapes): the four-wave tile kernel (gru_tile64.hip) against the twelve-wave one
(gru_pipe_fwd_kernel, layer by layer) and the per-sequence inference chain; build_memory alone, and Hpmn.forward_inference."""
import os, sys, time, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hpmn_amd import ops
dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["c3", "c2"]):
    c = dict(bench.CONFIGS[cfg]); c["config_id"] = cfg
    m = bench.build_model(c, tempfile.mkdtemp(), dev, seed=0)
    for rows in [int(x) for x in os.environ.get("ROWS", "2000,4096,8192").split(",")]:
        nb = (rows + c["batch"] - 1) // c["batch"]
        ids = torch.cat([b[0] for b in bench.synth_batches(c, nb, c["batch"], 7, dev)], 0)[:rows]
        emb, w = m.params["Embedding/emb_mtx"], m._gru_weights()
        def run(tile64):
            def f():
                ops.TILE64 = tile64
                return ops.tiled_forward_inference(m.spec, ids, emb, w)
            return f
        legs = [("tile64 (4 waves)", run(True)), ("tile (12 waves)", run(False)),
                ("per-sequence", lambda: ops.scan_forward_inference(m.spec, ids, emb, w))]
        for name, fn in legs:
            for _ in range(2): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print("%s %5d rows  %-18s %.2f ms  %.0f k sequences/s" % (cfg, rows, name, dt * 1e3, rows / dt / 1e3), flush=True)
        ops.TILE64 = True
    del m
