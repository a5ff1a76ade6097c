#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "rank_epilogue_term_forward or dt_in_launch" > gpurun_out/r06_dbg_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|AssertionError|assert " gpurun_out/r06_dbg_pytest.log | head -40
