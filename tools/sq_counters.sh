#!/bin/sh
# SQ counters of every kernel of the training step (per launch averages): sh tools/sq_counters.sh OUT [config]
# One rocprofv3 --pmc pass per group of counters (--kernel-trace only, as the MI355X guide prescribes); digest: OUT/sq_summary.txt
export TMPDIR=/tmp
out=${1:-gpurun_out/sq}; cfg=${2:-c3}
mkdir -p $out
: > $out/sq_summary.txt
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  d=$out/p; rm -rf $d
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- \
      python bench.py --config $cfg --steps 4 --warmup 2 --no-eval --no-parity-gate --no-cpu-baseline --no-auc --no-roofline --no-side-legs --no-input-pipeline > /dev/null 2> $out/err.txt < /dev/null
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $out/sq_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hpmn::", "")[:52]
    acc[r["Counter_Name"]][k] += float(r["Counter_Value"]); n[r["Counter_Name"]][k] += 1
for ctr in acc:
    print("== %s (average per launch)" % ctr)
    for k, v in sorted(acc[ctr].items(), key=lambda kv: -kv[1])[:14]:
        print("   %-52s %16.0f  (%d launches)" % (k, v / n[ctr][k], n[ctr][k]))
PY
  rm -rf $d
done
cat $out/sq_summary.txt | head -120
