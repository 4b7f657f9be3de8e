import sys, os, tempfile
sys.path.insert(0, os.getcwd())
import torch, bench
c = dict(bench.CONFIGS["c3"]); dev = torch.device("cuda:0")
m = bench.build_model(c, tempfile.mkdtemp(), dev)
batches = bench.synth_batches(c, 4, c["batch"], 1, dev)
ids = torch.cat([b[0] for b in batches], 0)
for _ in range(4):
    m.forward_inference(ids)
torch.cuda.synchronize()
