#!/bin/bash
# Round 6, call 2: lr_mode 3 (dt formed by the backward-data launch) — kernel + layer tests, then a same-box A/B of the step.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "rank_epilogue or fused_lora_dropout or lora_drop_dt" > gpurun_out/r06_call2_pytest.log 2>&1
echo "pytest kernels rc=$?"; tail -5 gpurun_out/r06_call2_pytest.log
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unet_gpu.py -m gpu -q -x -p no:cacheprovider -k "dropout or default_train or toy or graph_replay" > gpurun_out/r06_call2_pytest2.log 2>&1
echo "pytest train rc=$?"; tail -5 gpurun_out/r06_call2_pytest2.log
for rnd in 1 2; do
  for f in 0 1; do
    T2V_LORA_DT_FUSE=$f timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-roofline --no-default-mode --no-host-timing 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('DT_FUSE=$f round $rnd ms/step', d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r06_dt_fuse_ab.txt
