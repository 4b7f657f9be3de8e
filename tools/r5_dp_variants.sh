#!/bin/bash
# the data-parallel step with one rank on RCCL in bench.py's own timed loop, under a few switches
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --one-rank-rccl ${MODE:-rows}"
run() { echo -n "$1: "; env $1 $B 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'],3))"; }
for v in "$@"; do run "$v"; done
