#!/bin/bash
# Round 6 record set on ONE box: the whole GPU suite, the driver's bench command, the steady-state kernel trace, the per-signature
# in-step table + the isolated timings of the same signatures, the fused temporal unit probe.
#   gpurun --timeout 3000 -- 'bash scripts/r06_final.sh final'
tag=${1:-final}
mkdir -p gpurun_out
bash scripts/r06_suite.sh $tag
T2V_BENCH_SHAPE_TABLE=gpurun_out/r06_shapes_$tag.txt timeout 900 python bench.py > gpurun_out/r06_bench_default_$tag.json 2> gpurun_out/r06_bench_default_$tag.err
echo "bench rc=$?"
grep '^{' gpurun_out/r06_bench_default_$tag.json | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
r=d.get('roofline') or {}
print('ms/step', d['ms_per_step'], 'videos/s', d['value'], 'eval_train ms', d['config'].get('eval_train_ms_per_step'), 'roofline frac', r.get('frac'), 'gemm ms', r.get('kernel_ms_per_step'), 'eps rel', d['config'].get('eps_mse_rel_err'))
ns=r.get('north_star_kernels') or {}
print('wgrad', ns.get('lora_factor_gradients'))
t=ns.get('temporal_fused_forward_unit') or {}
print('fused unit', t.get('frac_mfma_peak'), t.get('per_width'), t.get('sampling_unet_forward_ms'))
print('cpu', d.get('cpu_baseline'))"
tail -2 gpurun_out/r06_bench_default_$tag.err
bash scripts/profile_bench.sh r06_$tag > /dev/null 2>&1; head -8 gpurun_out/r06_${tag}_window.txt | cut -c1-140
timeout 600 python scripts/gemm_vs_library.py 20 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_gemm_vs_library_$tag.txt
python scripts/instep_vs_isolated.py gpurun_out/r06_shapes_$tag.txt gpurun_out/r06_gemm_vs_library_$tag.txt > gpurun_out/r06_gemm_instep_vs_isolated_$tag.txt; cat gpurun_out/r06_gemm_instep_vs_isolated_$tag.txt
timeout 300 python scripts/temporal_fused_probe.py c2 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_temporal_fused_probe_c2_$tag.txt; tail -7 gpurun_out/r06_temporal_fused_probe_c2_$tag.txt | cut -c1-250
