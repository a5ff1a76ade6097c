#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_lora_grads_gpu.py -m gpu -q -p no:cacheprovider --durations=6 -s -k "sampler_with_the_native or shipped_grids" > gpurun_out/r06_call6_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|relerr|default mode, grid|s call" gpurun_out/r06_call6_pytest.log | head -40
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r06_call6_kernels.log 2>&1
echo "kernels rc=$?"; tail -3 gpurun_out/r06_call6_kernels.log
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/r06_bench_slab2.json 2> gpurun_out/r06_bench_slab2.err
echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench_slab2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['north_star_kernels']['conv3d_3x1x1_small_M'])
PY
