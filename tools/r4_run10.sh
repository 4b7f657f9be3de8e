#!/bin/sh
export TMPDIR=/tmp
o=gpurun_out/r4j
mkdir -p $o
for m in 1 0; do HPMN_WGRAD_BF16=$m python tools/wgrad_error.py 32 128 2>&1 | grep "bf16=" | grep -E "d_wg|d_bc|time"; done
for m in 1 0; do HPMN_WGRAD_BF16=$m python tools/wgrad_error.py 128 128 2>&1 | grep "bf16=" | grep -E "T=1024 D=128 d_wc|time"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "h128 or (tiny_and_odd and 128) or xlong_c4" 2>&1 | tail -2
for m in 0 1 0 1; do
HPMN_WGRAD_BF16=$m timeout 600 python bench.py --config c4 --steps 60 --no-auc --no-roofline --no-eval --no-parity-gate --no-cpu-baseline > $o/bench_c4_bf$m.json 2> $o/bench_c4_bf$m.err
python -c "
import json
d=json.load(open('$o/bench_c4_bf$m.json')); print('c4 bf16=$m', round(d['ms_per_step'],4), round(d['value']))"
done
