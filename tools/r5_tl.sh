#!/bin/sh
# one step's kernel timeline for a config: sh tools/r5_tl.sh c1
export TMPDIR=/tmp
cfg=${1:-c1}
rm -rf /tmp/prof_$cfg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$cfg -- python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --steps 12 --warmup 4 --config $cfg > /dev/null 2>&1
python tools/step_timeline.py $(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1)
