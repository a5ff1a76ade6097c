#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "sampler" > gpurun_out/r06_call15_sampler.log 2>&1
echo "sampler rc=$?"; grep -E "relerr|passed|failed|Error|error" gpurun_out/r06_call15_sampler.log | head -20 | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_tf2.json 2> gpurun_out/r06_bench_tf2.err
echo "bench rc=$?"; tail -3 gpurun_out/r06_bench_tf2.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench_tf2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
t = d['roofline']['north_star_kernels'].get('temporal_fused_forward_unit')
print(t and t['sampling_unet_forward_ms'], t and t['frac_mfma_peak'])
PY
