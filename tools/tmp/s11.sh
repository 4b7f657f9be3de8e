export TMPDIR=/tmp
o=gpurun_out/s11
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $o/test_parity.txt 2>&1; echo "rc=$?" >> $o/test_parity.txt
tail -3 $o/test_parity.txt
SWEEP_CONFIG=c3 sh tools/env_sweep.sh $o/sweep_c3 "HPMN_WGRAD_TSPLIT=1 HPMN_ADAM_MARKED=0" "HPMN_WGRAD_TSPLIT=1" "HPMN_ADAM_MARKED=0" "-" "HPMN_WGRAD_TSPLIT=2" "HPMN_WGRAD_TSPLIT=4"
BENCH_ARGS="--config c3 --steps 20 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c3 > /dev/null 2>&1
rm -rf $o/c3/prof
