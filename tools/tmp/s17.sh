export TMPDIR=/tmp
o=gpurun_out/s17
mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q > $o/test_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $o/test_gpu.txt
tail -4 $o/test_gpu.txt
timeout 600 python bench.py --steps 50 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('$o/bench_default.json')); print(round(d['ms_per_step'],4), d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['parity_gate']['pass'])"
