#!/bin/sh
export TMPDIR=/tmp
o=gpurun_out/r4d
mkdir -p $o
timeout 1800 python -m pytest tests/test_gpu_parity.py -q -k "tiled or deterministic or reproducible or int64 or more_rows or h128 or odd_lengths or gradients_match" > $o/test_sel.txt 2>&1; echo "rc=$?" >> $o/test_sel.txt
timeout 600 python tools/tile_eval_time.py 500 2000 4000 6000 > $o/tile_eval.txt 2>&1
for det in 0 1; do
HPMN_DET_SCATTER=$det timeout 600 python bench.py --steps 200 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c3_det$det.json 2> $o/bench_c3_det$det.err
done
HPMN_DET_SCATTER=0 timeout 600 python bench.py --steps 200 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c3_det0b.json 2> $o/bench_c3_det0b.err
for w in 4 8; do
HPMN_SCAN128_WAVES=$w timeout 600 python bench.py --config c4 --batch 250 --steps 50 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c4_b250_w$w.json 2> $o/bench_c4_b250_w$w.err
HPMN_SCAN128_WAVES=$w timeout 600 python bench.py --config c4 --batch 63 --steps 50 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c4_b63_w$w.json 2> $o/bench_c4_b63_w$w.err
done
timeout 600 python bench.py --config c4 --steps 50 --no-auc --no-roofline --no-parity-gate --no-cpu-baseline > $o/bench_c4.json 2> $o/bench_c4.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$o/prof.err < /dev/null; cd $GRAFT_REPO_ROOT
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $o/kernel_stats.csv; rm -rf $o/prof
tail -8 $o/test_sel.txt; cat $o/tile_eval.txt
for f in bench_c3_det0 bench_c3_det1 bench_c3_det0b bench_c4_b250_w4 bench_c4_b250_w8 bench_c4_b63_w4 bench_c4_b63_w8 bench_c4; do python -c "
import json,sys
d=json.load(open('$o/$f.json')); print('$f', round(d['ms_per_step'],4), round(d['value']), d.get('eval_sequences_per_s'))"; done
head -30 $o/kernel_stats.csv | cut -c1-150
