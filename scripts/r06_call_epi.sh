#!/bin/bash
# epilogue-operand rework of gemm_w8: kernel tests that cover it, then a same-box A/B against the previous library
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "w8 or gemm or conv or lora or lowrank or groupnorm or resnet or linear" > gpurun_out/epi_pytest.log 2>&1
tail -5 gpurun_out/epi_pytest.log
bash scripts/ab_bench.sh build_ab/libt2v_old.so 2 2>&1 | tee gpurun_out/epi_ab.txt
