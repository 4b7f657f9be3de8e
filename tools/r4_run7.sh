#!/bin/sh
export TMPDIR=/tmp
o=gpurun_out/r4g
mkdir -p $o
timeout 3000 python -m pytest tests -m gpu -q > $o/test_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $o/test_gpu.txt
tail -6 $o/test_gpu.txt
sh tools/collect_profiles.sh $o/profile_set c3 > $o/collect.txt 2>&1; tail -3 $o/collect.txt | cut -c1-400
for cfg in c1 c2 c4; do timeout 900 python bench.py --config $cfg > $o/bench_$cfg.json 2> $o/bench_$cfg.err; python -c "
import json
d=json.load(open('$o/bench_$cfg.json')); print('$cfg', round(d['ms_per_step'],4), round(d['value']), d.get('eval_sequences_per_s'), (d.get('eval_pass') or {}).get('sequences_per_s'), d['cpu_baseline']['value'])"; done
python -c "
import json
d=json.load(open('$o/profile_set/bench.json')); print('c3', round(d['ms_per_step'],4), round(d['value']), d.get('eval_sequences_per_s'), d.get('eval_pass'), d['cpu_baseline']['value'], d['roofline']['frac'], d['xlong_cadence'])"
cp profiles/r04_pmc_summary.json $o/ 2>/dev/null
