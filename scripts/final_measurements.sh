#!/bin/bash
# The round's record set on ONE box (one gpurun call): default bench line, steady-state kernel trace, library yardstick,
# C1/C3/C4/C5/dropout lines with their rooflines.  Outputs: gpurun_out/<tag>_*.json (copy into profiles/).
#   bash scripts/final_measurements.sh [tag]
tag=${1:-r03}
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
bash scripts/profile_bench.sh ${tag}_prof > /dev/null 2>&1
python bench.py --config c1 --no-cpu-baseline --steps 50 > gpurun_out/${tag}_bench_c1.json 2>/dev/null
python bench.py --config c3 --no-cpu-baseline --steps 20 > gpurun_out/${tag}_bench_c3.json 2>/dev/null
python bench.py --config c4 --no-cpu-baseline --steps 10 > gpurun_out/${tag}_bench_c4.json 2>/dev/null
python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/${tag}_bench_c5.json 2>/dev/null
python bench.py --config c5 --grad-checkpointing --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > gpurun_out/${tag}_bench_c5_ckpt.json 2>/dev/null
python bench.py --dropout --no-cpu-baseline --steps 30 > gpurun_out/${tag}_bench_dropout.json 2>/dev/null
python scripts/gemm_vs_library.py > gpurun_out/${tag}_gemm_vs_library.txt 2>&1
for f in default c1 c3 c4 c5 c5_ckpt dropout; do python -c "
import json,sys
d=json.loads(open('gpurun_out/${tag}_bench_$f.json').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$f', d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'), r.get('frac'), (r.get('secondary') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), d['config'].get('eps_mse_rel_err'), d['config'].get('default_mode_ms_per_step'))
"; done
head -12 gpurun_out/${tag}_prof_window.txt | cut -c1-140
