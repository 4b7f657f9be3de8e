export TMPDIR=/tmp; cd /root/repo
cat > /tmp/plan_only.py <<'PY'
import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from hpmn_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ids = torch.as_tensor(rng.integers(39002, 3308019, size=(500, 1001, 2)).astype(np.int32)).to(dev)
ids[:, :, 0] = ids[:, :1, 0] % 20000 + 19002
for _ in range(5):
    p = ops.ScatterPlan(ids, 16, want_rows=True, V=3308019)
torch.cuda.synchronize()
PY
rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python /tmp/plan_only.py > /dev/null 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
