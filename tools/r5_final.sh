#!/bin/sh
# final validation + profile set of round 5 (run from the repo root on the GPU box)
export TMPDIR=/tmp
o=gpurun_out/r5final
mkdir -p $o
if [ "$1" != "noprof" ]; then
sh tools/collect_profiles.sh $o/profile_set c3 > $o/collect.txt 2>&1; tail -2 $o/collect.txt | cut -c1-300
fi
for cfg in c1 c2 c4; do timeout 900 python bench.py --config $cfg > $o/bench_$cfg.json 2> $o/bench_$cfg.err; python -c "
import json
d=json.load(open('$o/bench_$cfg.json')); print('$cfg', round(d['ms_per_step'],4), round(d['value']), d.get('ms_per_step_all_fp32'), d.get('eval_sequences_per_s'), (d.get('eval_pass') or {}).get('sequences_per_s'), d['cpu_baseline']['value'], (d.get('data_parallel') or {}).get('one_rank_rccl_ms'))"; done
python -c "
import json
d=json.load(open('$o/profile_set/bench.json')); print('c3', round(d['ms_per_step'],4), round(d['value']), d.get('ms_per_step_all_fp32'), d.get('eval_sequences_per_s'), (d.get('eval_pass') or {}).get('sequences_per_s'), d['cpu_baseline']['value'], d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline']['traffic'], d['data_parallel'].get('one_rank_rccl_ms'), d['input_pipeline']['ratio_to_kernel_only'])"
for cfg in c1 c2 c4; do CFG=$cfg; rm -rf /tmp/prof_$cfg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$cfg -- python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --steps 12 --warmup 4 --config $cfg > /dev/null 2>&1; python tools/step_timeline.py $(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1) > $o/step_timeline_$cfg.txt; tail -1 $o/step_timeline_$cfg.txt; done
timeout 3000 python -m pytest tests -m gpu -q -rs > $o/test_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $o/test_gpu.txt
tail -4 $o/test_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
