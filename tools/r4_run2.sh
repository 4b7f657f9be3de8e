#!/bin/sh
# round-4 second GPU pass: whole GPU suite on ABI v10 (int64 ids), gather line-size experiment, gather PMC, bench x2
export TMPDIR=/tmp
o=gpurun_out/r4b
mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q > $o/test_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $o/test_gpu.txt
./tools/micro/gather_line > $o/gather_line.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/gl -- ./tools/micro/gather_line > /dev/null 2> $o/gl.err < /dev/null
f=$(find $o/gl -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/pmc_summary.py $f FETCH_SIZE > $o/gather_line_fetch.txt 2>&1
rm -rf $o/gl
python tools/gather_pmc.py time > $o/gather_time.json 2> $o/gather_time.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/gpmc -- python tools/gather_pmc.py count > /dev/null 2> $o/gpmc.err < /dev/null
f=$(find $o/gpmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python tools/gather_pmc.py digest $f $o/gather_time.json $o/gather_pmc.json > $o/gather_pmc.txt 2>&1
rm -rf $o/gpmc
timeout 900 python bench.py > $o/bench_c3.json 2> $o/bench_c3.err
timeout 600 python bench.py --steps 50 --no-auc --no-roofline --no-eval --no-parity-gate > $o/bench_c3_b.json 2> $o/bench_c3_b.err
tail -5 $o/test_gpu.txt; cat $o/gather_line.txt $o/gather_line_fetch.txt; grep "cpu baseline" $o/bench_c3.err $o/bench_c3_b.err
