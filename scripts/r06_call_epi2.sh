#!/bin/bash
mkdir -p gpurun_out
T2V_W8_DBG=8 timeout 600 python scripts/w8_epilogue_timeline.py 32768 2>&1 | grep "^M=" | tee gpurun_out/epi_timeline.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "w8 or gemm or conv or lora or lowrank or groupnorm or resnet or linear" > gpurun_out/epi_pytest.log 2>&1
tail -2 gpurun_out/epi_pytest.log
bash scripts/ab_bench.sh build_ab/libt2v_old.so 2 2>&1 | tee gpurun_out/epi_ab.txt
