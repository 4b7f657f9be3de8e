#!/bin/bash
out=gpurun_out/r6_b11; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$CFG $EXTRA $tag', round(json.loads(l)['ms_per_step'],4))"; }
CFG=c2; EXTRA=""
run default X=1; run det HPMN_DET_SCATTER=1; run compact HPMN_TABLE_GRAD=compact; run default_b X=1; run det_b HPMN_DET_SCATTER=1
CFG=c3; EXTRA="--id-law zipf"
run default X=1; run det HPMN_DET_SCATTER=1; run compact HPMN_TABLE_GRAD=compact
CFG=c1; EXTRA=""
run default X=1; run det HPMN_DET_SCATTER=1
