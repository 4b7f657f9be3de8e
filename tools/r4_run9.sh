#!/bin/sh
export TMPDIR=/tmp
o=gpurun_out/r4i
mkdir -p $o
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 200 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/$name.json 2> $o/$name.err; python -c "
import json
d=json.load(open('$o/$name.json')); print('$name', round(d['ms_per_step'],4), round(d['value']))"; }
run base HPMN_X=0
run pairbwd1 HPMN_PAIR_BWD=1
run pairfwd1 HPMN_PAIR_FWD=1
run pairboth HPMN_PAIR_BWD=1 HPMN_PAIR_FWD=1
run pairboth_earlybwd HPMN_PAIR_BWD=1 HPMN_PAIR_FWD=1 HPMN_EARLY_PASS=bwd
run base2 HPMN_X=0
run l0cut1 HPMN_L0_CUT=1
run l0cut2 HPMN_L0_CUT=2
