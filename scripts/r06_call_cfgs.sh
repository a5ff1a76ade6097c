#!/bin/bash
# the other BASELINE.json configurations on the final tree (default train mode): C1, C3, C4, C5 (no checkpointing / checkpointing)
mkdir -p gpurun_out
for cfg in c1 c3 c4 c5; do
  steps=10; [ $cfg = c5 ] && steps=5
  timeout 900 python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_$cfg.json 2> gpurun_out/r06_bench_$cfg.err
  echo "$cfg rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r06_bench_$cfg.json').read().strip().splitlines()[-1])
    print('$cfg', d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'))
except Exception as e:
    print('$cfg no line', e)
PY
done
timeout 900 python bench.py --config c5 --grad-checkpointing --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r06_bench_c5_ckpt.json 2> gpurun_out/r06_bench_c5_ckpt.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r06_bench_c5_ckpt.json').read().strip().splitlines()[-1])
print('c5 ckpt', d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'))
PY
