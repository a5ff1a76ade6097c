#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "groupnorm or gn" > gpurun_out/r06_call26_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r06_call26_pytest.log | cut -c1-200
bash scripts/ab_bench.sh build_ab/libt2v_old.so 2 2>&1 | tee gpurun_out/r06_gn_apply_ab.txt
for w in old new; do python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${w}_2.json").read().strip().splitlines()[-1])
print("$w", d["roofline"]["north_star_kernels"]["groupnorm_fwd(stats+apply)"])
PY
done 2>&1 | tee -a gpurun_out/r06_gn_apply_ab.txt
