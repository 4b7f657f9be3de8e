export TMPDIR=/tmp
o=gpurun_out/s12
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $o/test_parity.txt 2>&1; echo "rc=$?" >> $o/test_parity.txt
tail -3 $o/test_parity.txt
for c in c1 c2 c3; do SWEEP_CONFIG=$c sh tools/env_sweep.sh $o/sweep_$c "-"; done
BENCH_ARGS="--config c3 --steps 20 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c3 > /dev/null 2>&1
BENCH_ARGS="--config c1 --steps 20 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c1 > /dev/null 2>&1
rm -rf $o/*/prof
grep read_fwd_bwd $o/c3/timeline.txt $o/c1/timeline.txt
