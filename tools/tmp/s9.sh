export TMPDIR=/tmp
o=gpurun_out/s9
mkdir -p $o
BENCH_ARGS="--config c1 --steps 20 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c1 > /dev/null 2>&1
BENCH_ARGS="--config c2 --steps 20 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c2 > /dev/null 2>&1
rm -rf $o/*/prof
