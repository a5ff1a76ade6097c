#!/bin/bash
# Add the signatures of the reference's default train mode (LoRA dropout on: launches with a rank-wide epilogue term, T2VGemm.lr_mode)
# to the shipped tile table WITHOUT re-tuning what is in it: live tuning only times unknown signatures.
#   gpurun -- 'bash scripts/tune_lr_table.sh'      -> gpurun_out/gemm_tune_gfx950.txt (copy it over the shipped file)
set -u
T=text-to-video-finetuning_amd/gemm_tune_gfx950.txt
mkdir -p gpurun_out
for args in "--config c2 --dropout" "--config c1 --dropout"; do
  echo "== tuning: $args" >&2
  T2V_GEMM_AUTOTUNE=live timeout 900 python bench.py $args --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-default-mode --export-tune-table $T \
      > gpurun_out/tunelr_$(echo $args | tr -d ' -').json 2> gpurun_out/tunelr_$(echo $args | tr -d ' -').err || echo "   (failed: $args)" >&2
  wc -l $T >&2
done
cp $T gpurun_out/gemm_tune_gfx950.txt
