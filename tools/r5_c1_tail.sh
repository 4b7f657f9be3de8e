#!/bin/bash
# step time per config on the current library (no side legs)
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline"
for cfg in ${CFGS:-c1 c2 c3}; do for i in 1 2; do echo -n "$cfg: "; $B --config $cfg 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'],4))"; done; done
