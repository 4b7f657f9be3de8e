export TMPDIR=/tmp; cd "$(dirname "$0")/.."
rm -rf /tmp/pe; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -- python tools/eval128_time.py > /dev/null 2>&1
f=$(find /tmp/pe -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tile128" in r["Kernel_Name"] or "input_proj" in r["Kernel_Name"] or "gather_seq" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last tile pass at 4096 rows: the last 7 tile launches and the projections between them
idx = [i for i, r in enumerate(rows) if "tile128" in r["Kernel_Name"]]
lo = idx[-21]                     # three passes back: skip the per-sequence runs in between (they use input_proj too)
seq = rows[lo - 1:]
n = 0
for r in seq:
    if "tile128" in r["Kernel_Name"]: n += 1
    if n > 7: break
    print("%-60s %8.1f us" % (r["Kernel_Name"].replace("void hpmn::", "")[:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
