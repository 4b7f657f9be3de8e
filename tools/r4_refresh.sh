export TMPDIR=/tmp
o=gpurun_out/r4refresh
mkdir -p $o
sh tools/collect_profiles.sh $o/profile_set c3 > $o/collect.txt 2>&1; tail -2 $o/collect.txt | cut -c1-200
for cfg in c1 c2 c4; do timeout 900 python bench.py --config $cfg > $o/bench_$cfg.json 2> $o/bench_$cfg.err; done
python - <<'PY'
import json
for cfg,f in (('c3','gpurun_out/r4refresh/profile_set/bench.json'),('c1','gpurun_out/r4refresh/bench_c1.json'),('c2','gpurun_out/r4refresh/bench_c2.json'),('c4','gpurun_out/r4refresh/bench_c4.json')):
    d=json.load(open(f)); print(cfg, round(d['ms_per_step'],4), round(d['value']), round(d.get('eval_sequences_per_s') or 0), round((d.get('eval_pass') or {}).get('sequences_per_s',0)), round(d['cpu_baseline']['value'],1), d['cpu_baseline']['cores'], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('ms_per_launch'))
PY
cp profiles/r04_pmc_summary.json $o/
