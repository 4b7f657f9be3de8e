#!/bin/sh
export TMPDIR=/tmp
o=gpurun_out/r4e
mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "tiled or deterministic or reproducible or int64 or h128_b300 or gather" > $o/test_sel.txt 2>&1; echo "rc=$?" >> $o/test_sel.txt
for det in 0 1 0 1; do
HPMN_DET_SCATTER=$det timeout 600 python bench.py --steps 200 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c3_det$det.json 2> $o/bench_c3_det$det.err
python -c "
import json
d=json.load(open('$o/bench_c3_det$det.json')); print('det$det', round(d['ms_per_step'],4), round(d['value']))"
done
cd /tmp && HPMN_DET_SCATTER=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$o/prof.err < /dev/null; cd $GRAFT_REPO_ROOT
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $o/kernel_stats.csv; rm -rf $o/prof
tail -4 $o/test_sel.txt
grep -E "segsum|scatter|merge|radix|sort" $o/kernel_stats.csv | cut -c1-60,200-400 | head; grep -E "segsum|scatter_plan|embed_grad" $o/kernel_stats.csv | awk -F, '{print $1,$2,$4}' | cut -c1-120
