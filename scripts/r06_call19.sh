#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_lora_grads_gpu.py -m gpu -q -p no:cacheprovider -x -k "wgrad or lora or toy" > gpurun_out/r06_call19_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r06_call19_pytest.log | cut -c1-300
bash scripts/ab_bench.sh build_ab/libt2v_old.so 2 2>&1 | tee gpurun_out/r06_wgrad_ab.txt
for w in old new; do python - <<PY
import json
d = json.loads(open('gpurun_out/ab_${w}_2.json').read().strip().splitlines()[-1])
print('$w', d['roofline']['north_star_kernels']['lora_factor_gradients'])
PY
done 2>&1 | tee -a gpurun_out/r06_wgrad_ab.txt
