#!/bin/bash
# r6: three bf16 planes / six products (fp32-equivalent) against the r5 two-plane arithmetic and the fp32 kernels:
# error of the weight gradients against float64, gradient parity cases, ms/step at C3 / C4 / C2.
out=gpurun_out/r6_planes; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline"
for pl in 3 2; do
  HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 32 64 2>&1 | sed "s/^/planes=$pl /" >> $out/wgrad_error.txt
  HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 64 64 2>&1 | sed "s/^/planes=$pl /" >> $out/wgrad_error.txt
  HPMN_WGRAD_PLANES=$pl python tools/wgrad_error.py 128 128 2>&1 | sed "s/^/planes=$pl /" >> $out/wgrad_error.txt
done
HPMN_WGRAD_BF16=0 python tools/wgrad_error.py 32 64 2>&1 | sed "s/^/fp32 /" >> $out/wgrad_error.txt
cat $out/wgrad_error.txt
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],4))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for cfg in ${CFGS:-c3 c4 c2}; do
  for rep in 1 2; do
    $B --config $cfg > $out/${cfg}_p3_$rep.json 2> $out/${cfg}_p3_$rep.err; line $out/${cfg}_p3_$rep.json "$cfg planes3 rep$rep"
    HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2 HPMN_PROJ_PLANES=2 $B --config $cfg > $out/${cfg}_p2_$rep.json 2> $out/${cfg}_p2_$rep.err; line $out/${cfg}_p2_$rep.json "$cfg planes2 rep$rep"
  done
  HPMN_WGRAD_PLANES=2 $B --config $cfg > $out/${cfg}_w2.json 2> $out/${cfg}_w2.err; line $out/${cfg}_w2.json "$cfg wgrad2-only"
  HPMN_DX_PLANES=2 $B --config $cfg > $out/${cfg}_d2.json 2> $out/${cfg}_d2.err; line $out/${cfg}_d2.json "$cfg dx2-only"
  HPMN_WGRAD_BF16=0 HPMN_BWD_DX_INLOOP=0 HPMN_PROJ_BF16=0 HPMN_DX_BF16=0 HPMN_READ_BF16=0 $B --config $cfg > $out/${cfg}_fp32.json 2> $out/${cfg}_fp32.err; line $out/${cfg}_fp32.json "$cfg all-fp32"
done
HPMN_FUSED_SCATTER=2 $B --config c3 > $out/c3_fs2.json 2> $out/c3_fs2.err; line $out/c3_fs2.json "c3 fused-scatter-inloop"
python tools/r6_grad_planes.py 2>&1 | tee $out/grad_planes.txt
