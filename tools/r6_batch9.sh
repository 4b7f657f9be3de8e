#!/bin/bash
out=gpurun_out/r6_b9; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$CFG $EXTRA $tag', round(json.loads(l)['ms_per_step'],4))"; }
CFG=c3; EXTRA=""
run plain X=1; run plain_pf2 HPMN_WGRAD_PF=2; run plain_auxprio HPMN_AUX_PRIORITY=1; run plain_pf2_b HPMN_WGRAD_PF=2; run plain_b X=1
run p2 HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2; run p2_pf2 HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2 HPMN_WGRAD_PF=2
for pl in 3 2; do
  HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 32 64 2>&1 | grep "time B" | sed "s/^/planes=$pl pf1 /"
  HPMN_WGRAD_PF=2 HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 32 64 2>&1 | grep "time B\|d_wg" | sed "s/^/planes=$pl pf2 /" | cut -c1-140
  HPMN_WGRAD_PF=2 HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 64 64 2>&1 | grep "time B" | sed "s/^/planes=$pl pf2 /"
done
CFG=c2; run plain X=1; run pf2 HPMN_WGRAD_PF=2; run auxprio HPMN_AUX_PRIORITY=1
CFG=c3
EXTRA="--one-rank-rccl rows"; run q5 X=1; run q4 HPMN_ONE_RANK_QUEUES=4; run q5_auxprio HPMN_AUX_PRIORITY=1; run q4_auxprio HPMN_AUX_PRIORITY=1 HPMN_ONE_RANK_QUEUES=4
EXTRA="--one-rank-rccl allreduce"; run q5 X=1; run q4 HPMN_ONE_RANK_QUEUES=4; run q4_auxprio HPMN_AUX_PRIORITY=1 HPMN_ONE_RANK_QUEUES=4
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $out/test_gpu.txt 2>&1; tail -15 $out/test_gpu.txt
