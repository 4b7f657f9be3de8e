#!/bin/bash
# kernel timelines of one step: the plain step (compact gradient rows / dense gradient table) and the data-parallel step with one rank on RCCL
export TMPDIR=/tmp
out=gpurun_out/r5tl; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --steps 12 --warmup 4 --config ${CFG:-c3}"
for mode in auto dense; do
  rm -rf /tmp/prof_$mode
  HPMN_TABLE_GRAD=$mode timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$mode -- $B > $out/${mode}.out 2> $out/${mode}.err
  python tools/step_timeline.py $(find /tmp/prof_$mode -name "*kernel_trace.csv" | head -1) > $out/${CFG:-c3}_step_$mode.txt
done
rm -rf /tmp/prof_dp
DP1_MODES=rows DP1_STEPS=12 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dp -- python tools/dp_one_rank.py > $out/dp.out 2> $out/dp.err
python tools/step_timeline.py $(find /tmp/prof_dp -name "*kernel_trace.csv" | head -1) > $out/dp_rows_step.txt
