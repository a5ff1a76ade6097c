#!/bin/bash
mkdir -p gpurun_out
bash scripts/pmc_step.sh r06_pmc_step 2>&1 | tail -14
python scripts/pmc_summary.py gpurun_out/r06_pmc_step gpurun_out/r06_pmc_step.json | head -12
timeout 600 python scripts/determinism.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_determinism.txt; cat gpurun_out/r06_determinism.txt
