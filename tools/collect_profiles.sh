#!/bin/sh
# Round profile set on the GPU box: default bench line, rocprofv3 kernel stats + step timeline, and the two
# PMC passes (separate runs, --kernel-trace only, as the MI355X guide prescribes) digested into
# pmc_summary.json (stamped with the kernel-source sha).  $1 = output directory, $2 = config (default c3).
out=${1:-gpurun_out/profile_set}
cfg=${2:-c3}
export TMPDIR=/tmp
mkdir -p $out
BENCH_ARGS="--config $cfg --steps 20 --warmup 3" sh tools/profile_step.sh $out/step > /dev/null 2>&1
cp $out/step/kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
cp $out/step/timeline.txt $out/timeline.txt 2>/dev/null
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -- \
      python bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-eval --no-side-legs --no-input-pipeline --no-batch-sweep \
      > /dev/null 2> $out/pmc_$ctr.err < /dev/null
  f=$(find $out/pmc_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f $ctr > $out/pmc_$ctr.txt && cp $f $out/pmc_$ctr.csv
  rm -rf $out/pmc_$ctr
done
batch=$(python -c "import bench; print(bench.CONFIGS['$cfg']['batch'])")
python tools/pmc_digest.py $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv 6 $cfg $batch $out/pmc_summary.json
rm -f $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv
# bench.py reads the digest from profiles/ (and checks its source sha against the kernels it is about to run)
[ "$cfg" = "c3" ] && cp $out/pmc_summary.json profiles/${ROUND:-r06}_pmc_summary.json
rm -rf $out/step/prof
timeout 900 python bench.py --config $cfg > $out/bench.json 2> $out/bench.err < /dev/null
echo "bench rc=$?"
cut -c1-600 $out/bench.json
