#!/bin/bash
# housekeeping (gradient clear, the read path's weight gradients) on the auxiliary stream for small tables too: HPMN_AUX_MIN_NUMEL=0
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline"
for cfg in ${CFGS:-c1}; do for v in 0 16777216 0 16777216; do echo -n "$cfg HPMN_AUX_MIN_NUMEL=$v: "; HPMN_AUX_MIN_NUMEL=$v $B --config $cfg 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'],4))"; done; done
