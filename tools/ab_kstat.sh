#!/bin/sh
# A/B of per-kernel averages in the C3 step: sh tools/ab_kstat.sh OUT "ENV_A" "ENV_B"   ("-" = defaults)
o=$1; mkdir -p $o
for tag in A B; do
  if [ $tag = A ]; then e="$2"; else e="$3"; fi
  [ "$e" = "-" ] && e=""
  env $e sh tools/kstat.sh $o/$tag python bench.py --config ${SWEEP_CONFIG:-c3} --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-eval > $o/$tag.txt 2>&1
  echo "== $tag: $e"; grep -i "scan_bwd_feed\|pair_bwd\|wgrad_bf16\|scatter\|adam_table\|fwd_mfma\|pair_fwd\|read_fwd" $o/$tag.txt | cut -c1-60,90-140
done
