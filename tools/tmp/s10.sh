export TMPDIR=/tmp
o=gpurun_out/s10
mkdir -p $o
for T in 2 16 32 64 128 256 512; do
  echo "T=$T K=1 B=500: $(timeout 120 python tools/fwd_time.py 500 $T 1 2 2>&1 | tail -2 | tr '\n' ' ')" | tee -a $o/intercept.txt
done
for T in 16 64 256; do
  echo "T=$T K=1 B=128: $(timeout 120 python tools/fwd_time.py 128 $T 1 2 2>&1 | tail -2 | tr '\n' ' ')" | tee -a $o/intercept.txt
done
