#!/bin/sh
# Round profile set on the GPU box: default bench line, rocprofv3 kernel stats + step timeline, and the two
# PMC passes (separate runs, --kernel-trace only, as the MI355X guide prescribes).  $1 = output directory.
out=${1:-gpurun_out/profile_set}
export TMPDIR=/tmp
mkdir -p $out
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
echo "bench rc=$?"
sh tools/profile_step.sh $out/step > /dev/null 2>&1
cp $out/step/kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
cp $out/step/timeline.txt $out/timeline.txt 2>/dev/null
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -- \
      python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-auc > /dev/null 2> $out/pmc_$ctr.err < /dev/null
  f=$(find $out/pmc_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f $ctr > $out/pmc_$ctr.txt
  rm -rf $out/pmc_$ctr
done
rm -rf $out/step/prof
cut -c1-400 $out/bench.json
