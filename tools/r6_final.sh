#!/bin/sh
# final validation + profile set of round 6 (run from the repo root on the GPU box; one collection per round)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
o=gpurun_out/r6final
mkdir -p $o
ROUND=r06 sh tools/collect_profiles.sh $o/profile_set c3 > $o/collect.txt 2>&1; tail -2 $o/collect.txt | cut -c1-300
for cfg in c1 c2 c4; do timeout 900 python bench.py --config $cfg > $o/bench_$cfg.json 2> $o/bench_$cfg.err; python -c "
import json
d=json.load(open('$o/bench_$cfg.json')); print('$cfg', round(d['ms_per_step'],4), round(d['value']), 'two planes', d.get('ms_per_step_two_planes'), 'all fp32', d.get('ms_per_step_all_fp32'), 'eval', d.get('eval_sequences_per_s'), (d.get('eval_pass') or {}).get('sequences_per_s'), 'cpu', d['cpu_baseline']['value'], (d.get('data_parallel') or {}).get('one_rank_rccl_ms'), 'sweep', [(x.get('batch'), round(x.get('ms_per_step', 0), 3)) for x in d.get('batch_sweep', [])])"; done
python -c "
import json
d=json.load(open('$o/profile_set/bench.json')); print('c3', round(d['ms_per_step'],4), round(d['value']), 'two planes', d.get('ms_per_step_two_planes'), 'all fp32', d.get('ms_per_step_all_fp32'), 'eval', d.get('eval_sequences_per_s'), (d.get('eval_pass') or {}).get('sequences_per_s'), 'cpu', d['cpu_baseline']['value'], 'roofline', d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline']['traffic'], d['data_parallel'].get('one_rank_rccl_ms'), d['input_pipeline']['ratio_to_kernel_only'], 'sweep', [(x.get('batch'), round(x.get('ms_per_step', 0), 3)) for x in d.get('batch_sweep', [])])"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_c3_driver_cmd.json 2> $o/bench_c3_driver_cmd.err; python -c "
import json
d=json.load(open('$o/bench_c3_driver_cmd.json')); print('driver cmd: c3', round(d['ms_per_step'],4), round(d['value']))"
for cfg in c1 c2 c4; do rm -rf /tmp/prof_$cfg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$cfg -- python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep --steps 12 --warmup 4 --config $cfg > /dev/null 2>&1; python tools/step_timeline.py $(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1) > $o/step_timeline_$cfg.txt; tail -1 $o/step_timeline_$cfg.txt; done
rm -rf /tmp/prof_rows; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rows -- python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep --steps 12 --warmup 4 --config c3 --one-rank-rccl rows > /dev/null 2>&1; python tools/step_timeline.py $(find /tmp/prof_rows -name "*kernel_trace.csv" | head -1) > $o/step_timeline_dp_one_rank_rows.txt; tail -1 $o/step_timeline_dp_one_rank_rows.txt
python bench.py --gpus 2 --backend gloo --config c3 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep > $o/bench_c3_gloo2.json 2> $o/bench_c3_gloo2.err; grep '^{' $o/bench_c3_gloo2.json | python -c "
import json, sys
d = json.loads(sys.stdin.readline()); e = d.get('eval_pass', {})
print('gloo x2 on one GPU: eval_pass', e.get('sequences_per_s'), 'best', e.get('sequences_per_s_best'), 'replicas identical', d.get('replicas_identical'))"
for oc in 1 0; do HPMN_ONE_CALL_STEP=$oc python tools/host_enqueue_time.py c1 2>&1 | tail -1 | sed "s/^/one_call=$oc /"; done | tee $o/host_enqueue_c1.txt
timeout 3000 python -m pytest tests -m gpu -q -rs --durations=25 > $o/test_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $o/test_gpu.txt
tail -4 $o/test_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
