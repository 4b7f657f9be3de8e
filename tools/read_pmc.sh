#!/bin/sh
# SQ counters of read_fwd_bwd_kernel (sum over its launches / launches): sh tools/read_pmc.sh c1 "CTR1 CTR2 ..."  (a pass per group of <= 4)
export TMPDIR=/tmp
cfg=${1:-c1}; shift
out=gpurun_out/rpmc; mkdir -p $out
for grp in "$@"; do
  d=$out/p; rm -rf $d
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- \
      python bench.py --config $cfg --steps 4 --warmup 2 --no-eval --no-parity-gate --no-cpu-baseline --no-auc --no-roofline > /dev/null 2> $out/err.txt < /dev/null
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "read_fwd_bwd" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc: print("%-28s %14.0f per launch (%d launches)" % (k, acc[k] / max(n[k], 1), n[k]))
PY
done
