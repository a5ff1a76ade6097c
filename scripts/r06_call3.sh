#!/bin/bash
# Round 6, call 3: where lr_mode 3 spends its time — per-signature tables of the instrumented pass with and without it + kernel trace
mkdir -p gpurun_out
for f in 0 1; do
  T2V_LORA_DT_FUSE=$f T2V_BENCH_SHAPE_TABLE=gpurun_out/r06_shapes_dtfuse$f.txt timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-default-mode --no-host-timing > gpurun_out/r06_bench_dtfuse$f.json 2>/dev/null
done
T2V_LORA_DT_FUSE=1 bash scripts/profile_bench.sh r06_dtfuse1 > /dev/null 2>&1; head -40 gpurun_out/r06_dtfuse1_window.txt | cut -c1-150
