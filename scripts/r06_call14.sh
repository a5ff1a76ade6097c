#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "temporal_unit_fused or temporal_block_no_grad" > gpurun_out/r06_call14_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r06_call14_pytest.log | cut -c1-300
timeout 300 python scripts/temporal_fused_probe.py c2 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_temporal_fused_probe_c2_v4.txt; tail -7 gpurun_out/r06_temporal_fused_probe_c2_v4.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_tf.json 2> gpurun_out/r06_bench_tf.err
echo "bench rc=$?"; tail -3 gpurun_out/r06_bench_tf.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench_tf.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['roofline']['north_star_kernels'].get('temporal_fused_forward_unit'), indent=1))
PY
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -x -k "sampler or vae_decode" > gpurun_out/r06_call14_sampler.log 2>&1
echo "sampler rc=$?"; tail -3 gpurun_out/r06_call14_sampler.log | cut -c1-300
