#!/bin/bash
# Re-tune the shipped GEMM tile table (text-to-video-finetuning_amd/gemm_tune_gfx950.txt) on the GPU box: every configuration of
# the bench is run once with T2V_GEMM_AUTOTUNE=live (unknown signatures are timed on first use, candidates = the 4-wave tiles of
# gemm.hip and the 8-wave configurations of gemm_w8.hip); each run starts from the table the previous one exported.
#   gpurun -- 'bash scripts/tune_gemm_table.sh'      -> gpurun_out/gemm_tune_gfx950.txt (copy it over the shipped file)
set -u
T=text-to-video-finetuning_amd/gemm_tune_gfx950.txt
mkdir -p gpurun_out
: > $T
for args in "--config c2" "--config c2 --dropout" "--config c1" "--config c4" "--config c3" "--config c5 --grad-checkpointing"; do
  echo "== tuning: $args" >&2
  T2V_GEMM_AUTOTUNE=live timeout 900 python bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --export-tune-table $T \
      > gpurun_out/tune_$(echo $args | tr -d ' -').json 2> gpurun_out/tune_$(echo $args | tr -d ' -').err || echo "   (failed: $args)" >&2
  wc -l $T >&2
done
cp $T gpurun_out/gemm_tune_gfx950.txt
