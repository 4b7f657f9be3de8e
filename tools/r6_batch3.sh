#!/bin/bash
# r6 batch 3: the pinned, software-pipelined in-loop input gradient; wgrad with its own operand block in registers; C4 with the
# lo weight plane of the H = 128 projection in LDS; helper-stream priority
out=gpurun_out/r6_b3; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gradients_match or three_training or c_abi_alone or one_call or input_gradient_formed or scatter or wrong_labels or layer0_backward_split" > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],4))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run() { tag=$1; shift; env "$@" $B --config $CFG > $out/${CFG}_$tag.json 2> $out/${CFG}_$tag.err; line $out/${CFG}_$tag.json "$CFG $tag"; }
CFG=c3
for rep in 1 2; do
  run p3_$rep X=1
  run p2_$rep HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2
done
run w2dx3 HPMN_WGRAD_PLANES=2
run w3dx2 HPMN_DX_PLANES=2
run p3_occ2 HPMN_WGRAD_OCC=2
run p2_occ2 HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2 HPMN_WGRAD_OCC=2
run p3_prio HPMN_SIDE_PRIORITY=1
run p3_fs2 HPMN_FUSED_SCATTER=2
run fp32 HPMN_WGRAD_BF16=0 HPMN_BWD_DX_INLOOP=0 HPMN_PROJ_BF16=0 HPMN_DX_BF16=0 HPMN_READ_BF16=0
run p3_dxepi HPMN_BWD_DX_INLOOP=0
for pl in 3 2; do
  HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 32 64 2>&1 | grep "time B" | sed "s/^/planes=$pl occ3 /"
  HPMN_WGRAD_OCC=2 HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 32 64 2>&1 | grep "time B" | sed "s/^/planes=$pl occ2 /"
  HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 64 64 2>&1 | grep "time B" | sed "s/^/planes=$pl /"
  HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 128 128 2>&1 | grep "time B" | sed "s/^/planes=$pl /"
done
CFG=c4
run p3 X=1
run p2 HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2 HPMN_PROJ_PLANES=2
run p3_proj2 HPMN_PROJ_PLANES=2
run p3_dx2 HPMN_DX_PLANES=2
run p3_w2 HPMN_WGRAD_PLANES=2
run p3_prio HPMN_SIDE_PRIORITY=1
run p2_prio HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2 HPMN_PROJ_PLANES=2 HPMN_SIDE_PRIORITY=1
CFG=c2
run p3 X=1
run p2 HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2
python tools/r6_grad_planes.py 2>&1 | tee $out/grad_planes.txt
