export TMPDIR=/tmp
o=gpurun_out/s5
mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "h128 or 128" > $o/test_h128.txt 2>&1; echo "rc=$?" >> $o/test_h128.txt
tail -3 $o/test_h128.txt
SWEEP_CONFIG=c4 sh tools/env_sweep.sh $o/sweep "HPMN_WGRAD_ALTERNATE=0" "-" 
BENCH_ARGS="--config c4 --steps 10 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c4 > /dev/null 2>&1
rm -rf $o/c4/prof
