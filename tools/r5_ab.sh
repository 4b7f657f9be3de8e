#!/bin/bash
# A/B of the compact-gradient-row step against the r4 dense-gradient-table step, per config (bench lines without the side legs)
out=gpurun_out/r5ab; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline"
for cfg in ${CFGS:-c3 c2 c4 c1}; do
  for mode in auto dense; do
    HPMN_TABLE_GRAD=$mode $B --config $cfg > $out/${cfg}_$mode.json 2> $out/${cfg}_$mode.err
    python - <<PY
import json
d=json.loads([l for l in open("$out/${cfg}_$mode.json") if l.startswith("{")][0])
print("$cfg", "$mode", "ms_per_step", round(d["ms_per_step"],4))
PY
  done
done
