#!/bin/bash
out=gpurun_out/r6_b12; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scatter or gradients_match or embed or one_call or c_abi" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$CFG $EXTRA $tag', round(d['ms_per_step'],4), d['config'].get('table_scatter'))"; }
CFG=c2; EXTRA=""
run auto X=1; run plain HPMN_SCATTER_HOT=0; run hot HPMN_SCATTER_HOT=1; run auto_b X=1; run plain_b HPMN_SCATTER_HOT=0
CFG=c1; run auto X=1; run plain HPMN_SCATTER_HOT=0; run hot HPMN_SCATTER_HOT=1
CFG=c3; run auto X=1; run hot HPMN_SCATTER_HOT=1
EXTRA="--id-law zipf"; run auto X=1; run atomic_plain HPMN_DET_SCATTER=0 HPMN_SCATTER_HOT=0; run atomic_hot HPMN_DET_SCATTER=0 HPMN_SCATTER_HOT=1; run det HPMN_DET_SCATTER=1
CFG=c4; run auto X=1; run atomic_hot HPMN_DET_SCATTER=0 HPMN_SCATTER_HOT=1
