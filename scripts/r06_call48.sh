#!/bin/bash
mkdir -p gpurun_out
bash scripts/r06_suite.sh kmajor
python bench.py --config c3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r06_bench_c3.json 2>/dev/null
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_c2_kmajor.json 2>/dev/null
for f in r06_bench_c3 r06_bench_c2_kmajor; do python - <<PY
import json
d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
s=d["roofline"]["secondary"]
print("$f: ms/step", d["ms_per_step"], "videos/s", d["value"], "K-major family ms", s["kernel_ms_per_step"], "frac", s["frac"], "gemm frac", d["roofline"]["frac"])
PY
done
