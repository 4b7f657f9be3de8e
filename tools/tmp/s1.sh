export TMPDIR=/tmp
o=gpurun_out/s1
mkdir -p $o
timeout 900 python -m pytest tests -m gpu -q -x > $o/test_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $o/test_gpu.txt
tail -3 $o/test_gpu.txt
BENCH_ARGS="--config c3 --steps 20 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c3 > /dev/null 2>&1
HPMN_DET_SCATTER=1 BENCH_ARGS="--config c3 --steps 20 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c3det > /dev/null 2>&1
BENCH_ARGS="--config c1 --steps 20 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c1 > /dev/null 2>&1
BENCH_ARGS="--config c4 --steps 10 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c4 > /dev/null 2>&1
rm -rf $o/*/prof
sh tools/env_sweep.sh $o/sweep "-" "HPMN_DET_SCATTER=1"
