export TMPDIR=/tmp
o=gpurun_out/s13
mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "h128 or 128" > $o/test_h128.txt 2>&1; echo "rc=$?" >> $o/test_h128.txt
tail -3 $o/test_h128.txt
SWEEP_CONFIG=c4 sh tools/env_sweep.sh $o/sweep "-" "HPMN_LIB_PATH=hpmn_amd/lib/variants/libhpmn_pfb3.so" "HPMN_LIB_PATH=hpmn_amd/lib/variants/libhpmn_pfb4.so"
BENCH_ARGS="--config c4 --steps 10 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c4 > /dev/null 2>&1
rm -rf $o/c4/prof
timeout 600 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-auc --no-parity-gate > $o/bench_c4.json 2>$o/bench_c4.err
python -c "
import json
d=json.load(open('$o/bench_c4.json')); print('c4', round(d['ms_per_step'],4), d.get('eval_sequences_per_s'), (d.get('eval_pass') or {}).get('sequences_per_s'))"
