#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_lora_grads_gpu.py -m gpu -q -p no:cacheprovider --durations=6 -s -k "sampler_with_the_native or shipped_grids" > gpurun_out/r06_call5_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|relerr|default mode, grid|s call" gpurun_out/r06_call5_pytest.log | head -40
