#!/bin/bash
# A/B of the training read launch: split-bf16 fragments (r5, default) against the r4 fp32 launch (HPMN_READ_BF16=0), per config
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline"
for cfg in ${CFGS:-c3 c1 c2 c4}; do
  for mode in ${MODES:-1 0 1 0}; do
    echo -n "$cfg HPMN_READ_BF16=$mode: "
    HPMN_READ_BF16=$mode $B --config $cfg 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'],4))"
  done
done
