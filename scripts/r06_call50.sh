#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "layernorm or norm or attention or geglu" > gpurun_out/r06_call50_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r06_call50_pytest.log | cut -c1-200
bash scripts/ab_bench.sh build_ab/libt2v_old.so 2 2>&1 | tee gpurun_out/r06_ln_ab.txt
for w in old new; do python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${w}_2.json").read().strip().splitlines()[-1])
ns = d["roofline"]["north_star_kernels"]
print("$w", "ln_fwd", ns["layernorm_fwd"]["ms_per_step"], ns["layernorm_fwd"]["frac_hbm_peak"], "ln_bwd", ns["layernorm_bwd"]["ms_per_step"], ns["layernorm_bwd"]["frac_hbm_peak"])
PY
done 2>&1 | tee -a gpurun_out/r06_ln_ab.txt
