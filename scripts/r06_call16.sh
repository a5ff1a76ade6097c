#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_lora_grads_gpu.py tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -s -k "sampler or no_grad_forward or temporal_block_no_grad" > gpurun_out/r06_call16_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "relerr|passed|failed|Error|error|folded" gpurun_out/r06_call16_pytest.log | head -30 | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_tf3.json 2> gpurun_out/r06_bench_tf3.err
echo "bench rc=$?"; tail -3 gpurun_out/r06_bench_tf3.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench_tf3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
t = d['roofline']['north_star_kernels'].get('temporal_fused_forward_unit')
print(t and t['sampling_unet_forward_ms'], t and t['frac_mfma_peak'], t and t['one_launch_vs_separate_relerr'])
PY
