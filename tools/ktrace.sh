#!/bin/sh
# kernel-by-kernel timeline of the LAST repetition of a command's kernels.  $1 = out dir, $2 = number of trailing launches, rest = command
out=$1; n=$2; shift; shift
export TMPDIR=/tmp
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof -- "$@" > $out/stdout.txt 2> $out/err.txt < /dev/null
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python - "$t" "$n" <<'PY' | tee $out/trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "hpmn" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-70s start %8.1f  dur %7.1f us" % (r["Kernel_Name"].replace("void hpmn::", "").replace("hpmn::", "")[:70], (s - t0) / 1e3, (e - s) / 1e3))
PY
rm -rf $out/prof
