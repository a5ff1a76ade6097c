#!/bin/bash
# Round 6, call 4: lr_mode 3 with keep-bit planes — kernel + layer tests, A/B of the step (0: off, 1: where the estimate says it pays, 2: everywhere)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "rank_epilogue or fused_lora_dropout or lora_drop_dt" > gpurun_out/r06_call4_pytest.log 2>&1
echo "pytest kernels rc=$?"; tail -5 gpurun_out/r06_call4_pytest.log
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unet_gpu.py -m gpu -q -x -p no:cacheprovider -k "dropout or default_train or toy or graph_replay or checkpoint" > gpurun_out/r06_call4_pytest2.log 2>&1
echo "pytest train rc=$?"; tail -5 gpurun_out/r06_call4_pytest2.log
for rnd in 1 2; do
  for f in 0 1 2; do
    T2V_LORA_DT_FUSE=$f timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-roofline --no-default-mode --no-host-timing 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('DT_FUSE=$f round $rnd ms/step', d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r06_dt_fuse_ab2.txt
for f in 0 2; do
  T2V_LORA_DT_FUSE=$f T2V_BENCH_SHAPE_TABLE=gpurun_out/r06_shapes_plane_dtfuse$f.txt timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-default-mode --no-host-timing > /dev/null 2>&1
done
