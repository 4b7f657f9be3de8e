#!/bin/sh
export TMPDIR=/tmp
o=gpurun_out/r4f
mkdir -p $o
python tools/segsum_time.py 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "deterministic or reproducible" > $o/test_sel.txt 2>&1; echo "rc=$?" >> $o/test_sel.txt; tail -2 $o/test_sel.txt
for det in 0 1 0 1; do
HPMN_DET_SCATTER=$det timeout 600 python bench.py --steps 200 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c3_det$det.json 2> $o/bench_c3_det$det.err
python -c "
import json
d=json.load(open('$o/bench_c3_det$det.json')); print('det$det', round(d['ms_per_step'],4), round(d['value']))"
done
cd /tmp && HPMN_DET_SCATTER=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$o/prof.err < /dev/null; cd $GRAFT_REPO_ROOT
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $o/kernel_stats.csv; rm -rf $o/prof
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r4f/kernel_stats.csv')):
    n=r['Name']
    if any(k in n for k in ('embed_grad_scatter','adam_table_kernel','gru_wgrad_kernel<2, 1','segsum','gru_scan_bwd_feed','gru_fwd_mfma_kernel<32')):
        print('  %-60s calls %4s avg %9.1f us' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
