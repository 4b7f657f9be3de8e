#!/bin/bash
out=gpurun_out/r6_b10; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$CFG $EXTRA $tag', round(json.loads(l)['ms_per_step'],4))"; }
CFG=c3; EXTRA=""
run plain X=1; run plain_auxprio HPMN_AUX_PRIORITY=1
EXTRA="--one-rank-rccl rows"; run q5 X=1; run q4 HPMN_ONE_RANK_QUEUES=4; run q5_auxprio HPMN_AUX_PRIORITY=1; run q4_auxprio HPMN_AUX_PRIORITY=1 HPMN_ONE_RANK_QUEUES=4
EXTRA="--one-rank-rccl allreduce"; run q5_auxprio HPMN_AUX_PRIORITY=1; run q4_auxprio HPMN_AUX_PRIORITY=1 HPMN_ONE_RANK_QUEUES=4
CFG=c2; EXTRA=""; run plain X=1; run auxprio HPMN_AUX_PRIORITY=1
CFG=c4; run plain X=1; run auxprio HPMN_AUX_PRIORITY=1
