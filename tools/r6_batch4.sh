#!/bin/bash
# r6 batch 4: where does the three-plane in-loop input gradient's +0.22 ms go?  per-kernel timelines of one step
out=gpurun_out/r6_b4; mkdir -p $out
export TMPDIR=/tmp
tl() { tag=$1; shift; rm -rf /tmp/prof_$tag; env "$@" timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep --steps 12 --warmup 4 --config ${CFG:-c3} > /dev/null 2>$out/tl_$tag.err; python tools/step_timeline.py $(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1) > $out/timeline_${CFG:-c3}_$tag.txt; echo "== $tag"; grep -E "scan_bwd_feed|step:|wgrad_bf16_kernel<2, 1|embed_grad_scatter|adam_table_kernel<1>" $out/timeline_${CFG:-c3}_$tag.txt | cut -c1-110; }
tl p2 HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2
tl w2dx3 HPMN_WGRAD_PLANES=2
tl p3 X=1
tl p3_dxepi HPMN_BWD_DX_INLOOP=0
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_call or split_gradient_kernels_are" > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt; grep "max |grad" $out/pytest.txt
python tools/r6_grad_planes.py 2>&1 | tee $out/grad_planes.txt
for oc in 1 0; do HPMN_ONE_CALL_STEP=$oc python tools/host_enqueue_time.py c1 2>&1 | tail -1 | sed "s/^/one_call=$oc /"; done
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
for sg in 0 1; do HPMN_DP_SIDE_GROUP=$sg $B --config c3 --one-rank-rccl rows 2>$out/rows_sg$sg.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('one-rank rows side_group=$sg', round(json.loads(l)['ms_per_step'],4))"; done
$B --config c3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('plain', round(json.loads(l)['ms_per_step'],4))"
python -m pytest tests/test_gpu_dp.py -x -q -m gpu -k "prepared_a_step_ahead or every_collective or dataset_sharded" > $out/pytest_dp.txt 2>&1; tail -3 $out/pytest_dp.txt
