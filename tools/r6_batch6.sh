#!/bin/bash
out=gpurun_out/r6_b6; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/r6_dp_eval_dbg.py 2>&1 | grep "^rank" 
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
for sg in 0 1; do rm -rf /tmp/prof_sg$sg; HPMN_DP_SIDE_GROUP=$sg timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sg$sg -- $B --steps 12 --warmup 4 --config c3 --one-rank-rccl rows > /dev/null 2>$out/sg$sg.err; python tools/step_timeline.py $(find /tmp/prof_sg$sg -name "*kernel_trace.csv" | head -1) > $out/timeline_rows_sg$sg.txt; echo "== side group $sg"; cut -c1-110 $out/timeline_rows_sg$sg.txt; done
