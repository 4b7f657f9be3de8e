#!/bin/bash
out=gpurun_out/r6_b5; mkdir -p $out
python tools/r6_eval_dbg.py 2>&1 | tail -12
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$CFG $tag', round(json.loads(l)['ms_per_step'],4))"; }
CFG=c3
for rep in 1 2; do run p3_$rep X=1; run p2_$rep HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2; done
run w2dx3 HPMN_WGRAD_PLANES=2
run w3dx2 HPMN_DX_PLANES=2
run fp32 HPMN_WGRAD_BF16=0 HPMN_BWD_DX_INLOOP=0 HPMN_PROJ_BF16=0 HPMN_DX_BF16=0 HPMN_READ_BF16=0
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gradients_match or input_gradient_formed or scatter or one_call or split_gradient_kernels_are or layer0_backward_split" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt; grep "max |grad" $out/pytest.txt
export TMPDIR=/tmp
rm -rf /tmp/prof_p3; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_p3 -- $B --steps 12 --warmup 4 --config c3 > /dev/null 2>&1; python tools/step_timeline.py $(find /tmp/prof_p3 -name "*kernel_trace.csv" | head -1) > $out/timeline_c3_p3.txt; tail -12 $out/timeline_c3_p3.txt | cut -c1-110
CFG=c4
rm -rf /tmp/prof_c4; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c4 -- $B --steps 8 --warmup 3 --config c4 > /dev/null 2>&1; python tools/step_timeline.py $(find /tmp/prof_c4 -name "*kernel_trace.csv" | head -1) > $out/timeline_c4_p3.txt; cat $out/timeline_c4_p3.txt | cut -c1-110
rm -rf /tmp/prof_c4b; HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2 HPMN_PROJ_PLANES=2 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c4b -- $B --steps 8 --warmup 3 --config c4 > /dev/null 2>&1; python tools/step_timeline.py $(find /tmp/prof_c4b -name "*kernel_trace.csv" | head -1) > $out/timeline_c4_p2.txt; tail -3 $out/timeline_c4_p2.txt | cut -c1-110
