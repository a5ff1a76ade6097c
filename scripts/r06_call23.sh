#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_lora_grads_gpu.py tests/test_kernels_gpu.py tests/test_unet_gpu.py -m gpu -q -p no:cacheprovider -s -k "sampler or no_grad_forward or temporal_block_no_grad or unet_forward_matches" > gpurun_out/r06_call23_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "relerr|passed|failed|Error|error|folded" gpurun_out/r06_call23_pytest.log | head -30 | cut -c1-300
bash scripts/profile_cmd.sh r06_sampling_c2 scripts/sampling_probe.py c2 20 > /dev/null 2>&1; grep "CFG-pair" gpurun_out/r06_sampling_c2.log; head -32 gpurun_out/r06_sampling_c2_stats.txt | cut -c1-160
timeout 600 python scripts/sampling_probe.py c4 10 2>&1 | grep CFG-pair
