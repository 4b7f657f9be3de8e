#!/bin/bash
out=gpurun_out/r6_b17; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c_abi_alone or one_call or three_training or smoke" 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-auc --no-eval --no-roofline --no-parity-gate --no-side-legs --no-input-pipeline --no-batch-sweep"
run() { tag=$1; shift; env "$@" $B --config $CFG $EXTRA 2>$out/${CFG}_$tag.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$CFG $EXTRA $tag', round(d['ms_per_step'],4))"; }
CFG=c1; EXTRA=""
run aside X=1; run inline HPMN_READ_GRADS_ASIDE=0; run aside_b X=1; run inline_b HPMN_READ_GRADS_ASIDE=0
EXTRA="--batch 512"; run aside X=1; run inline HPMN_READ_GRADS_ASIDE=0
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
