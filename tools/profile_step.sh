#!/bin/sh
# rocprofv3 kernel trace + stats of the default bench; prints the last step's timeline.  $1 = output dir
out=${1:-gpurun_out/prof_step}
export TMPDIR=/tmp
mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python bench.py --no-cpu-baseline --no-auc --no-roofline --no-side-legs --no-input-pipeline --no-batch-sweep ${BENCH_ARGS} > $out/bench.json 2> $out/err.txt < /dev/null
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv
t=$(find $out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/step_timeline.py $t > $out/timeline.txt
find $out/prof -name "*kernel_trace.csv" -delete
cat $out/timeline.txt
cut -c1-160 $out/bench.json
