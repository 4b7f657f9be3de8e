#!/bin/bash
out=gpurun_out/r6_b8; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$CFG $EXTRA $tag', round(json.loads(l)['ms_per_step'],4))"; }
CFG=c3; EXTRA=""; run plain X=1
EXTRA="--one-rank-rccl rows"; run q5 X=1; run q4 HPMN_ONE_RANK_QUEUES=4
EXTRA="--one-rank-rccl allreduce"; run q5 X=1
rm -rf /tmp/prof_rows; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rows -- $B --steps 12 --warmup 4 --config c3 --one-rank-rccl rows > /dev/null 2>$out/rows_prof.err; python tools/step_timeline.py $(find /tmp/prof_rows -name "*kernel_trace.csv" | head -1) > $out/timeline_rows.txt; grep -v "rocprim\|fillBuffer\|at::native" $out/timeline_rows.txt | cut -c1-105
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > $out/test_gpu.txt 2>&1; tail -15 $out/test_gpu.txt
