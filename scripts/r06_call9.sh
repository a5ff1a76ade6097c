#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "temporal_unit_fused or temporal_block_no_grad" > gpurun_out/r06_call9_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r06_call9_pytest.log | cut -c1-300
T2V_TF_ABLATE=1 timeout 300 python scripts/temporal_fused_probe.py c2 > gpurun_out/r06_temporal_fused_probe_v2.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06_temporal_fused_probe_v2.txt | tail -12
