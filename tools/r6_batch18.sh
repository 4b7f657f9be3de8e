#!/bin/bash
# r6: re-measure r5's negatives under the round's new conditions (register ledger, prefetch-2 weight gradients)
out=gpurun_out/r6_b18; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$CFG $EXTRA $tag', round(d['ms_per_step'],4))"; }
CFG=c3; EXTRA=""
run base X=1
run tsplit2 HPMN_WGRAD_TSPLIT=2
run tsplit3 HPMN_WGRAD_TSPLIT=3
run l0cut1 HPMN_L0_CUT=1
run l0cut2 HPMN_L0_CUT=2
run pairs0 HPMN_PAIR_FWD=1 HPMN_PAIR_BWD=1
run pairfwd0 HPMN_PAIR_FWD=1
run earlybwd HPMN_EARLY_PASS=bwd
run base_b X=1
run compact HPMN_TABLE_GRAD=compact
run det HPMN_DET_SCATTER=1
