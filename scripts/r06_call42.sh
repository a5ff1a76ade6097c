#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_unet_gpu.py -m gpu -q -p no:cacheprovider -x -k "linear_fwd_bwd or conv2d or conv3d or full_finetune or full_backward or vae or stable_lora" > gpurun_out/r06_call42_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r06_call42_pytest.log | cut -c1-200
timeout 300 python scripts/kmajor_probe.py 2>&1 | grep tokens | tee gpurun_out/r06_kmajor_probe_after.txt
python bench.py --config c3 --steps 15 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_c3_after.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_c3_after.json").read().strip().splitlines()[-1])
s=d["roofline"]["secondary"]
print("C3 ms/step", d["ms_per_step"], "K-major family ms", s["kernel_ms_per_step"], "frac", s["frac"], "achieved", s["achieved"])
PY
