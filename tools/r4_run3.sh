#!/bin/sh
export TMPDIR=/tmp
o=gpurun_out/r4c
mkdir -p $o
timeout 3000 python -m pytest tests -m gpu -q > $o/test_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $o/test_gpu.txt
timeout 600 python tools/tile_eval_time.py 500 1000 2000 4000 8000 > $o/tile_eval.txt 2>&1
python tools/gather_pmc.py time > $o/gather_time.json 2> $o/gather_time.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/gpmc -- python tools/gather_pmc.py count > /dev/null 2> $o/gpmc.err < /dev/null
f=$(find $o/gpmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/gather_pmc.py digest $f $o/gather_time.json $o/gather_pmc.json > $o/gather_pmc.txt 2>&1
rm -rf $o/gpmc
timeout 900 python bench.py > $o/bench_c3.json 2> $o/bench_c3.err
timeout 600 python bench.py --steps 50 --no-auc --no-roofline --no-eval --no-parity-gate > $o/bench_c3_b.json 2> $o/bench_c3_b.err
HPMN_DET_SCATTER=0 timeout 600 python bench.py --steps 100 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c3_atomic.json 2> $o/bench_c3_atomic.err
timeout 600 python bench.py --config c1 --no-cpu-baseline --no-roofline > $o/bench_c1.json 2> $o/bench_c1.err
HPMN_DET_SCATTER=0 timeout 600 python bench.py --config c1 --no-cpu-baseline --no-roofline > $o/bench_c1_atomic.json 2> $o/bench_c1_atomic.err
tail -15 $o/test_gpu.txt; cat $o/tile_eval.txt; cat $o/gather_pmc.txt; grep "cpu baseline" $o/bench_c3.err $o/bench_c3_b.err
for f in bench_c3 bench_c3_b bench_c3_atomic bench_c1 bench_c1_atomic; do python -c "
import json,sys
d=json.load(open('$o/$f.json')); print('$f', round(d['ms_per_step'],4), round(d['value']))"; done
