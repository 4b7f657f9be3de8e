#!/bin/sh
export TMPDIR=/tmp
o=gpurun_out/r4h
mkdir -p $o
for m in 0 1 0 1; do
HPMN_WGRAD_BF16=$m timeout 600 python bench.py --steps 200 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c3_bf$m.json 2> $o/bench_c3_bf$m.err
python -c "
import json
d=json.load(open('$o/bench_c3_bf$m.json')); print('c3 bf16=$m', round(d['ms_per_step'],4), round(d['value']))"
done
for m in 0 1; do
HPMN_WGRAD_BF16=$m timeout 600 python bench.py --config c2 --steps 200 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c2_bf$m.json 2> $o/bench_c2_bf$m.err
python -c "
import json
d=json.load(open('$o/bench_c2_bf$m.json')); print('c2 bf16=$m', round(d['ms_per_step'],4), round(d['value']))"
done
cd /tmp && HPMN_WGRAD_BF16=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$o/prof.err < /dev/null; cd $GRAFT_REPO_ROOT
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $o/kernel_stats.csv
t=$(find $o/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/step_timeline.py $t > $o/timeline.txt; rm -rf $o/prof
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r4h/kernel_stats.csv')):
    n=r['Name']
    if any(k in n for k in ('embed_grad_scatter','adam_table_kernel','wgrad','gru_scan_bwd_feed','gru_pair_bwd')):
        print('  %-60s calls %4s avg %9.1f us' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
tail -22 $o/timeline.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "gradients_match or c_abi_alone or tiny_and_odd or three_training or wrong_labels or item_branch" 2>&1 | tail -3
