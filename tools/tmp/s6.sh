export TMPDIR=/tmp
o=gpurun_out/s6
mkdir -p $o
SWEEP_CONFIG=c4 sh tools/env_sweep.sh $o/sweep "-" "HPMN_ADAM_EARLY_WGS=512" "HPMN_ADAM_EARLY_WGS=2048" "HPMN_EARLY_SPLIT=0.5" "HPMN_DX_BF16=0" "HPMN_PROJ_BF16=0"
for b in 250 63; do
for m in 3 0; do
 v=$(HPMN_SCAN128_SOLO=$m timeout 300 python bench.py --config c4 --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-eval 2>>$o/err.txt | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
 echo "B=$b solo=$m: $v" | tee -a $o/batch.txt
done; done
