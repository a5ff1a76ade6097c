#!/bin/bash
# Round 5, GPU call 4: measurement artefacts of the current build — kernel statistics of the default-mode step (rocprofv3 kernel
# trace), HBM-side traffic (two --pmc passes, one-graph form), and the default-mode lines of configs C4 / C5.
mkdir -p gpurun_out
bash scripts/profile_bench.sh r05a
bash scripts/pmc_step.sh r05a_pmc
timeout 400 python bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-default-mode > gpurun_out/r05_bench_c4.json 2> gpurun_out/r05_bench_c4.err
echo "c4 rc=$?"; grep '^{' gpurun_out/r05_bench_c4.json | cut -c1-300; tail -2 gpurun_out/r05_bench_c4.err
timeout 500 python bench.py --config c5 --grad-checkpointing --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-default-mode > gpurun_out/r05_bench_c5_ckpt.json 2> gpurun_out/r05_bench_c5_ckpt.err
echo "c5 rc=$?"; grep '^{' gpurun_out/r05_bench_c5_ckpt.json | cut -c1-300; tail -2 gpurun_out/r05_bench_c5_ckpt.err
