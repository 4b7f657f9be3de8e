#!/bin/bash
# one rank on RCCL, rows exchange: the step with a stand-in wire (sleep kernels; 8 ranks at 300 GB/s; uniform ids and ~0.21 of
# the rows = the Zipf(1.1) volume) against the step without one and the plain step
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline"
for i in 1 2; do
echo -n "plain: "; $B 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'],4))"
for w in "" "300,8" "300,8,0.213" "150,8"; do echo -n "rows, HPMN_DP_WIRE_STANDIN='$w': "; HPMN_DP_WIRE_STANDIN=$w $B --one-rank-rccl rows 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'],4))"; done
done
