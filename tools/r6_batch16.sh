#!/bin/bash
out=gpurun_out/r6_b16; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$CFG $EXTRA $tag', round(d['ms_per_step'],4), d['config'].get('table_scatter','')[:60])"; }
CFG=c2; EXTRA=""
run single X=1; run paired HPMN_PAIR_SINGLE=0; run single_b X=1; run paired_b HPMN_PAIR_SINGLE=0
EXTRA="--batch 256"; run single X=1; run paired HPMN_PAIR_SINGLE=0
CFG=c3; EXTRA=""; run auto X=1; run hot1 HPMN_SCATTER_HOT_HINT=1; run hot0 HPMN_SCATTER_HOT_HINT=0; run auto_b X=1
EXTRA="--id-law zipf"; run auto X=1
EXTRA="--batch 128"; run single X=1; run paired HPMN_PAIR_SINGLE=0
timeout 2400 python -m pytest tests -m gpu -q -x > $out/test_gpu.txt 2>&1; tail -4 $out/test_gpu.txt
