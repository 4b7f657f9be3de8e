#!/bin/bash
out=gpurun_out/r6_b7; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp.py -x -q -m gpu -k "one_call or split_gradient_kernels_are or dataset_sharded or prepared_a_step_ahead or every_collective" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt; grep "max |grad" $out/pytest.txt
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$CFG $EXTRA $tag', round(json.loads(l)['ms_per_step'],4))"; }
CFG=c3; EXTRA=""
for rep in 1 2; do run prio1_$rep X=1; run prio0_$rep HPMN_SIDE_PRIORITY=0; done
run prio1_p2 HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2
CFG=c4; run prio1 X=1; run prio0 HPMN_SIDE_PRIORITY=0; run prio1_p2 HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2 HPMN_PROJ_PLANES=2
CFG=c2; run prio1 X=1; run prio0 HPMN_SIDE_PRIORITY=0
CFG=c1; run prio1 X=1; run prio0 HPMN_SIDE_PRIORITY=0
CFG=c3; EXTRA="--one-rank-rccl rows"
run sg0_prio1_q5 X=1
run sg0_prio1_q4 HPMN_ONE_RANK_QUEUES=4
run sg0_prio1_q8 HPMN_ONE_RANK_QUEUES=8
run sg0_prio0_q5 HPMN_SIDE_PRIORITY=0
run sg1_prio1_q5 HPMN_DP_SIDE_GROUP=1
run sg1_prio1_q4 HPMN_DP_SIDE_GROUP=1 HPMN_ONE_RANK_QUEUES=4
EXTRA="--one-rank-rccl allreduce"
run ar_prio1_q5 X=1
run ar_prio1_q4 HPMN_ONE_RANK_QUEUES=4
