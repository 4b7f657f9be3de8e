#!/bin/bash
# kernel-by-kernel breakdown of one Hpmn.eval pass at the config's shape (rocprofv3 kernel trace of tools/eval_pass_time.py-like run)
export TMPDIR=/tmp; cd "$(dirname "$0")/.."
cat > /tmp/evalpass.py <<'PY'
import os, sys, tempfile, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
cfg = os.environ.get("CFG", "c3")
dev = torch.device("cuda:0")
c = dict(bench.CONFIGS[cfg]); c["config_id"] = cfg
m = bench.build_model(c, tempfile.mkdtemp(), dev, seed=0)
ids = torch.cat([b[0] for b in bench.synth_batches(c, 16, c["batch"], 7, dev)], 0)
ds = dict(ids=ids.cpu().numpy(), label=np.random.default_rng(1).integers(0, 2, size=ids.shape[0]).astype(np.int32))
for _ in range(3):
    m.eval(ds, 4 * c["batch"])
torch.cuda.synchronize()
PY
rm -rf /tmp/pev; rocprofv3 --kernel-trace --output-format csv -d /tmp/pev -- python /tmp/evalpass.py > /dev/null 2>&1
python - "$(find /tmp/pev -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last third of the launches = the last eval call
n = len(rows) // 3
last = rows[-n:]
t = collections.OrderedDict()
for r in last:
    k = r["Kernel_Name"].replace("void hpmn::", "").replace("hpmn::", "").split("(")[0][:50]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    t.setdefault(k, [0, 0.0]); t[k][0] += 1; t[k][1] += d
tot = sum(v[1] for v in t.values())
for k, (cnt, d) in sorted(t.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-52s x%3d %9.1f us  %5.1f %%" % (k, cnt, d, 100 * d / tot))
print("sum of kernel time %.1f us; wall of the call %.1f us" % (tot, (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3))
PY
