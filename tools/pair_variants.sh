#!/bin/sh
# forward of the C3 shape under each variant library x SIMD assignment: durations of the pair launches.  args: variant names
for v in "$@"; do
  for m in 1 2; do
    lib=hpmn_amd/lib/variants/libhpmn_$v.so
    [ "$v" = "default" ] && lib=hpmn_amd/lib/libhpmn_hip.so
    FWD_ONLY=1 HPMN_LIB_PATH=$lib HPMN_PAIR_FWD=$m sh tools/ktrace.sh gpurun_out/pv 6 python tools/fwd_time.py > /dev/null 2>&1
    echo "$v mode$m: $(grep pair_fwd gpurun_out/pv/trace.txt | awk '{printf "%s ", $(NF-1)}') total $(tail -1 gpurun_out/pv/stdout.txt | sed 's/.*forward \([0-9.]*\) us.*/\1/')"
  done
done
