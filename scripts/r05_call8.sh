#!/bin/bash
# Round 5, last GPU call: smoke() of the entry point + the other BASELINE.json rows on the final tree (default train mode).
mkdir -p gpurun_out
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -2
for cfg in c1 c3; do
  timeout 400 python bench.py --config $cfg --steps 20 --warmup 2 --no-cpu-baseline --no-default-mode > gpurun_out/r05_bench_$cfg.json 2> gpurun_out/r05_bench_$cfg.err
  echo "$cfg rc=$?"; grep '^{' gpurun_out/r05_bench_$cfg.json | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d.get('roofline') or {}
print(d['config']['workload'][:60], 'ms/step', d['ms_per_step'], 'videos/s', d['value'], 'frac', r.get('frac'), 'secondary', (r.get('secondary') or {}).get('achieved'), 'peak GB', d['config']['peak_hbm_gb'])"
done
timeout 300 python bench.py --eval-train --steps 50 --warmup 3 --no-cpu-baseline --no-default-mode > gpurun_out/r05_bench_eval_train.json 2> gpurun_out/r05_bench_eval_train.err
echo "eval rc=$?"; grep '^{' gpurun_out/r05_bench_eval_train.json | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d.get('roofline') or {}
print('eval_train ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'gemm ms', r.get('kernel_ms_per_step'))"
