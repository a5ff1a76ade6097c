#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "attention" > gpurun_out/r06_call54_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r06_call54_pytest.log | cut -c1-200
bash scripts/ab_bench.sh build_ab/libt2v_old.so 2 2>&1 | tee gpurun_out/r06_attn_ab.txt
for w in old new; do python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${w}_2.json").read().strip().splitlines()[-1])
ns = d["roofline"]["north_star_kernels"]
print("$w", "spatial", ns["spatial_attention_core"]["ms_per_step"], ns["spatial_attention_core"]["frac_mfma_peak"], "temporal", ns["temporal_attention_core"]["ms_per_step"], "text", ns["text_cross_attention_core"]["ms_per_step"])
PY
done 2>&1 | tee -a gpurun_out/r06_attn_ab.txt
for w in old new; do
  if [ $w = old ]; then export T2V_LIB_FILE=$PWD/build_ab/libt2v_old.so; else unset T2V_LIB_FILE; fi
  python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/ab_c4_$w.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/ab_c4_$w.json').read().strip().splitlines()[-1]); print('$w C4 ms/step', d['ms_per_step'])" | tee -a gpurun_out/r06_attn_ab.txt
done
