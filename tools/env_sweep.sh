#!/bin/sh
# C3 step time under environment switches: sh tools/env_sweep.sh OUT "VAR=V VAR2=V" "..." ...  ("-" = the defaults).
# Three bench runs per setting (steps 60), ms_per_step of each.
export TMPDIR=/tmp
o=$1; shift
mkdir -p $o
cfg=${SWEEP_CONFIG:-c3}
for s in "$@"; do
  line="$s:"
  for r in 1 2 3; do
    if [ "$s" = "-" ]; then e=""; else e="$s"; fi
    v=$(env $e timeout 300 python bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-eval 2>>$o/err.txt | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
    line="$line $v"
  done
  echo "$line" | tee -a $o/sweep.txt
done
