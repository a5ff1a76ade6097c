#!/bin/bash
mkdir -p gpurun_out
for cfg in c4 c5; do
  timeout 900 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_$cfg.json 2> gpurun_out/r06_bench_$cfg.err
  echo "$cfg rc=$?"; tail -2 gpurun_out/r06_bench_$cfg.err | cut -c1-300
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r06_bench_$cfg.json').read().strip().splitlines()[-1])
    print('$cfg', d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'), d['config']['workload'][-160:])
except Exception as e:
    print('$cfg no line', e)
PY
done
timeout 900 python bench.py --config c5 --grad-checkpointing --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_c5_ckpt.json 2> gpurun_out/r06_bench_c5_ckpt.err
echo "c5 ckpt rc=$?"; python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_c5_ckpt.json').read().strip().splitlines()[-1])
print('c5 ckpt', d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'))
PY
timeout 600 python bench.py --config c1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_c1.json 2> gpurun_out/r06_bench_c1.err
timeout 600 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_c3.json 2> gpurun_out/r06_bench_c3.err
for cfg in c1 c3; do python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_$cfg.json').read().strip().splitlines()[-1])
print('$cfg', d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'))
PY
done
