python bench.py > gpurun_out/r2f_bench_default.json 2> gpurun_out/r2f_bench_default.err
bash scripts/profile_bench.sh r2f_prof > /dev/null 2>&1
python bench.py --config c1 --no-cpu-baseline --no-roofline > gpurun_out/r2f_bench_c1.json 2>/dev/null
python bench.py --config c3 --no-cpu-baseline --no-roofline --steps 5 > gpurun_out/r2f_bench_c3.json 2>/dev/null
python bench.py --config c4 --no-cpu-baseline --steps 5 > gpurun_out/r2f_bench_c4.json 2>/dev/null
python bench.py --config c5 --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > gpurun_out/r2f_bench_c5.json 2>/dev/null
python bench.py --dropout --no-cpu-baseline --no-roofline > gpurun_out/r2f_bench_dropout.json 2>/dev/null
for f in default c1 c3 c4 c5 dropout; do python -c "
import json,sys
d=json.loads(open('gpurun_out/r2f_bench_$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), d['config'].get('eps_mse_rel_err'))
"; done
head -12 gpurun_out/r2f_prof_window.txt | cut -c1-140
