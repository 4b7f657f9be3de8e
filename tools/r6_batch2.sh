#!/bin/bash
# r6 batch 2: the restructured in-loop input gradient (split shared per k slice), the one-call step, DP evaluation, workloads
out=gpurun_out/r6_b2; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp.py -x -q -m gpu -k "gradients_match or three_training or c_abi_alone or one_call or dataset_sharded or wrong_labels or two_rank_training_matches" > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],4))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
  $B --config c3 > $out/c3_p3_$rep.json 2> $out/c3_p3_$rep.err; line $out/c3_p3_$rep.json "c3 planes3 rep$rep"
  HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2 $B --config c3 > $out/c3_p2_$rep.json 2> $out/c3_p2_$rep.err; line $out/c3_p2_$rep.json "c3 planes2 rep$rep"
done
HPMN_WGRAD_PLANES=2 $B --config c3 > $out/c3_w2.json 2> $out/c3_w2.err; line $out/c3_w2.json "c3 wgrad2-only(dx3)"
HPMN_DX_PLANES=2 $B --config c3 > $out/c3_d2.json 2> $out/c3_d2.err; line $out/c3_d2.json "c3 dx2-only(wgrad3)"
HPMN_WGRAD_PLANES=2 HPMN_DX_PLANES=2 $B --config c2 > $out/c2_p2.json 2> $out/c2_p2.err; line $out/c2_p2.json "c2 planes2"
$B --config c2 > $out/c2_p3.json 2> $out/c2_p3.err; line $out/c2_p3.json "c2 planes3"
for oc in 1 0; do
  HPMN_ONE_CALL_STEP=$oc python tools/host_enqueue_time.py c1 2>&1 | tail -2 | sed "s/^/one_call=$oc /"
  HPMN_ONE_CALL_STEP=$oc $B --config c1 > $out/c1_oc$oc.json 2> $out/c1_oc$oc.err; line $out/c1_oc$oc.json "c1 one_call=$oc"
done
# one rank on RCCL, rows exchange: early exchange on the default communicator (r6 default) vs the second communicator (r5)
for sg in 0 1; do
  HPMN_DP_SIDE_GROUP=$sg $B --config c3 --one-rank-rccl rows > $out/c3_rows_sg$sg.json 2> $out/c3_rows_sg$sg.err; line $out/c3_rows_sg$sg.json "c3 one-rank rows side_group=$sg"
done
HPMN_BENCH_NEXT_IDS=0 $B --config c3 --one-rank-rccl rows > $out/c3_rows_nonext.json 2> $out/c3_rows_nonext.err; line $out/c3_rows_nonext.json "c3 one-rank rows no-next-ids"
# the full default lines of C1 / C2 on the specified workloads (batch sweep, eval legs)
for cfg in c1 c2; do
  timeout 900 python bench.py --config $cfg --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  python - $out/bench_$cfg.json $cfg <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[2], "ms", round(d["ms_per_step"],4), "all_fp32", d.get("ms_per_step_all_fp32"), "eval_pass", (d.get("eval_pass") or {}).get("sequences_per_s"), "sweep", [(x.get("batch"), round(x.get("ms_per_step",0),3), round(x.get("sequences_per_s",0))) for x in d.get("batch_sweep",[])])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
# DP eval leg: two gloo ranks sharing this GPU against the single process (eval_pass only matters)
python bench.py --gpus 2 --backend gloo --config c3 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep > $out/c3_gloo2.json 2> $out/c3_gloo2.err
python - $out/c3_gloo2.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); print("c3 gloo2 eval_pass", d.get("eval_pass"), "replicas", d.get("replicas_identical"))
except Exception as e: print("gloo2 FAILED", e)
PY
python bench.py --config c3 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep > $out/c3_single_eval.json 2> $out/c3_single_eval.err
python - $out/c3_single_eval.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); print("c3 single eval_pass", d.get("eval_pass"))
except Exception as e: print("single FAILED", e)
PY
