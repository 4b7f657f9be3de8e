#!/bin/bash
out=gpurun_out/r6_b15; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$CFG $EXTRA $tag', round(d['ms_per_step'],4), d.get('dp_phases_ms'))"; }
CFG=c3; EXTRA=""; run plain X=1
EXTRA="--one-rank-rccl rows"
run q5 X=1; run q4 HPMN_ONE_RANK_QUEUES=4; run q5_planaux HPMN_PLAN_ON_AUX=1; run q4_planaux HPMN_PLAN_ON_AUX=1 HPMN_ONE_RANK_QUEUES=4; run q6_planaux HPMN_PLAN_ON_AUX=1 HPMN_ONE_RANK_QUEUES=6; run q8_planaux HPMN_PLAN_ON_AUX=1 HPMN_ONE_RANK_QUEUES=8
python -m pytest tests/test_gpu_dp.py -x -q -m gpu -k "prepared_a_step_ahead or every_collective" 2>&1 | tail -2
HPMN_PLAN_ON_AUX=1 python -m pytest tests/test_gpu_dp.py -x -q -m gpu -k "prepared_a_step_ahead or every_collective" 2>&1 | tail -2
