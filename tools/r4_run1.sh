#!/bin/sh
# round-4 first GPU pass: new tests, gather PMC, bench (cpu_baseline reproducibility)
export TMPDIR=/tmp
o=gpurun_out/r4a
mkdir -p $o
(rocprofv3 -L 2>&1 | grep -i -E "TCC_EA0?_RDREQ|TCC_EA0?_RD|FETCH|TCC_HIT|TCC_MISS|TCC_REQ" | head -60) > $o/counters.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_dp.py -x -q > $o/test_dp.txt 2>&1; echo "dp rc=$?" >> $o/test_dp.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wrong_labels or b66" > $o/test_wrong.txt 2>&1; echo "rc=$?" >> $o/test_wrong.txt
python tools/gather_pmc.py time > $o/gather_time.json 2> $o/gather_time.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/gpmc -- python tools/gather_pmc.py count > /dev/null 2> $o/gpmc.err < /dev/null
f=$(find $o/gpmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/gather_pmc.py digest $f $o/gather_time.json $o/gather_pmc.json > $o/gather_pmc.txt 2>&1
rm -rf $o/gpmc
timeout 900 python bench.py > $o/bench_c3.json 2> $o/bench_c3.err
timeout 600 python bench.py --steps 50 --no-auc --no-roofline --no-eval --no-parity-gate > $o/bench_c3_b.json 2> $o/bench_c3_b.err
tail -3 $o/test_dp.txt; tail -3 $o/test_wrong.txt; cat $o/gather_pmc.txt; cut -c1-300 $o/bench_c3.json
