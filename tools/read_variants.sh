#!/bin/sh
# duration of read_fwd_bwd_kernel in a step under library variants: sh tools/read_variants.sh "v1 v2 ..." "c3 c1"
for v in $1; do
  lib=hpmn_amd/lib/variants/libhpmn_$v.so
  [ "$v" = "default" ] && lib=hpmn_amd/lib/libhpmn_hip.so
  for c in ${2:-c3 c1}; do
    HPMN_LIB_PATH=$lib sh tools/ktrace.sh gpurun_out/ra 60 python bench.py --config $c --steps 6 --warmup 2 --no-eval --no-parity-gate --no-cpu-baseline --no-auc --no-roofline > /dev/null 2>&1
    echo "$v $c: read_fwd_bwd $(grep read_fwd_bwd gpurun_out/ra/trace.txt | tail -1 | awk '{print $(NF-1)}') us  reduce $(grep read_reduce gpurun_out/ra/trace.txt | tail -1 | awk '{print $(NF-1)}') us"
  done
done
