export TMPDIR=/tmp
o=gpurun_out/s8
mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $o/test_parity.txt 2>&1; echo "rc=$?" >> $o/test_parity.txt
tail -3 $o/test_parity.txt
SWEEP_CONFIG=c4 sh tools/env_sweep.sh $o/sweep_c4 "HPMN_DX_LDS=0" "-" "HPMN_DX_LDS_GRID=512"
SWEEP_CONFIG=c3 sh tools/env_sweep.sh $o/sweep_c3 "-"
SWEEP_CONFIG=c2 sh tools/env_sweep.sh $o/sweep_c2 "-"
BENCH_ARGS="--config c4 --steps 10 --warmup 3 --no-parity-gate --no-eval" sh tools/profile_step.sh $o/c4 > /dev/null 2>&1
rm -rf $o/c4/prof
