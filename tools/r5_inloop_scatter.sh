#!/bin/bash
# the scatter inside the loop of layer 0's reverse scan (HPMN_FUSED_SCATTER=2) against the scatter launch behind it
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline"
for cfg in ${CFGS:-c3 c2}; do for v in 2 0 2 0; do echo -n "$cfg HPMN_FUSED_SCATTER=$v: "; HPMN_FUSED_SCATTER=$v $B --config $cfg 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'],4))"; done; done
