#!/bin/sh
# final validation + profile set of the round (run from the repo root on the GPU box)
export TMPDIR=/tmp
o=gpurun_out/r4final
mkdir -p $o
timeout 3000 python -m pytest tests -m gpu -q > $o/test_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $o/test_gpu.txt
tail -6 $o/test_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-eval > $o/bench_c3_dp2_gloo.json 2> $o/bench_c3_dp2_gloo.err; echo "dp2 rc=$?"; cut -c1-250 $o/bench_c3_dp2_gloo.json
sh tools/collect_profiles.sh $o/profile_set c3 > $o/collect.txt 2>&1; tail -2 $o/collect.txt | cut -c1-300
for cfg in c1 c2 c4; do timeout 900 python bench.py --config $cfg > $o/bench_$cfg.json 2> $o/bench_$cfg.err; python -c "
import json
d=json.load(open('$o/bench_$cfg.json')); print('$cfg', round(d['ms_per_step'],4), round(d['value']), d.get('eval_sequences_per_s'), (d.get('eval_pass') or {}).get('sequences_per_s'), d['cpu_baseline']['value'])"; done
python -c "
import json
d=json.load(open('$o/profile_set/bench.json')); print('c3', round(d['ms_per_step'],4), round(d['value']), d.get('eval_sequences_per_s'), (d.get('eval_pass') or {}).get('sequences_per_s'), d['cpu_baseline']['value'], d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline']['traffic'], d['xlong_cadence'])"
cp profiles/r04_pmc_summary.json $o/ 2>/dev/null
