#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -x -k "linear_fwd_bwd or conv2d or conv3d or full_finetune or kmajor or wgrad" > gpurun_out/r06_call38_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r06_call38_pytest.log | cut -c1-200
for k in 1 2; do for which in old new; do
  if [ $which = old ]; then export T2V_LIB_FILE=$PWD/build_ab/libt2v_old.so; else unset T2V_LIB_FILE; fi
  python bench.py --config c3 --steps 15 --warmup 3 --no-cpu-baseline > gpurun_out/ab_c3_${which}_$k.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_c3_${which}_$k.json").read().strip().splitlines()[-1])
s=d["roofline"]["secondary"]
print("$which $k: C3 ms/step", d["ms_per_step"], "K-major family ms", s["kernel_ms_per_step"], "frac", s["frac"])
PY
done; done 2>&1 | tee gpurun_out/r06_c3_splitk_raster_ab.txt
