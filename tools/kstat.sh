#!/bin/sh
# per-kernel average durations of a command under rocprofv3 --kernel-trace --stats.  $1 = out dir, rest = command
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- "$@" > $out/stdout.txt 2> $out/err.txt < /dev/null
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv
rm -rf $out/prof
python - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-90s calls %5s  avg %9.1f us  total %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
