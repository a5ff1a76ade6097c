#!/bin/bash
for w in 1600 800 560 400 280; do
  T2V_KMAJOR_WGS=$w python bench.py --config c3 --steps 15 --warmup 3 --no-cpu-baseline > gpurun_out/r06_c3_wgs_$w.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_c3_wgs_$w.json").read().strip().splitlines()[-1])
s=d["roofline"]["secondary"]
print("T2V_KMAJOR_WGS=$w: C3 ms/step", d["ms_per_step"], "K-major family ms", s["kernel_ms_per_step"], "frac", s["frac"])
PY
done
