#!/bin/bash
# Round 6, call 1: baseline of the round-5 tree on this round's box — the driver's bench command with the per-signature GEMM table
# of the instrumented pass, the steady-state kernel trace, and the same signatures in isolation beside the vendor library.
mkdir -p gpurun_out
T2V_BENCH_SHAPE_TABLE=gpurun_out/r06_shapes_base.txt timeout 900 python bench.py > gpurun_out/r06_bench_base.json 2> gpurun_out/r06_bench_base.err
echo "bench rc=$?"; grep '^{' gpurun_out/r06_bench_base.json | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('ms/step', d['ms_per_step'], 'eval_train ms', d['config'].get('eval_train_ms_per_step'), 'frac', (d.get('roofline') or {}).get('frac'))"
bash scripts/profile_bench.sh r06_base > /dev/null 2>&1; head -12 gpurun_out/r06_base_window.txt | cut -c1-150
timeout 600 python scripts/gemm_vs_library.py 30 > gpurun_out/r06_gemm_vs_library_base.txt 2>&1; tail -20 gpurun_out/r06_gemm_vs_library_base.txt
