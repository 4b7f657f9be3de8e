#!/bin/sh
# cycles of workgroup 0 of read_fwd_bwd_kernel per phase (READ_CLOCK build: python tools/variant.py rclk -DREAD_CLOCK); phase
# numbers are the RCLK marks in read_path.hip.   sh tools/read_clock.sh "c3 c1" [variant]
for c in ${1:-c3 c1}; do
  echo "== $c ${2:-rclk}"
  HPMN_LIB_PATH=hpmn_amd/lib/variants/libhpmn_${2:-rclk}.so timeout 200 python bench.py --config $c --steps 3 --warmup 2 --no-eval --no-parity-gate --no-cpu-baseline --no-auc --no-roofline --no-side-legs --no-input-pipeline 2>/dev/null | grep RCLK | tail -1
done
