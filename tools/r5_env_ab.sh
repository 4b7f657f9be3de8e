#!/bin/bash
# the plain C3 (or $CFG) step under a few environment switches, bench.py's timed loop, legs off
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --config ${CFG:-c3}"
run() { echo -n "$1: "; env $1 $B 2>/dev/null | python -c "import sys,json; print(round(json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'],4))"; }
for v in "$@"; do run "$v"; done
