#!/bin/bash
# Round 5, GPU call 6: the restructured register epilogue of the 8-wave GEMM family (bias from LDS, batched residual / rank-phase
# operands, stores back to back): kernel tests, same-box A/B against the library of the previous commit, counter-pass attempts.
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_kernels_gpu.py tests/test_lora_grads_gpu.py -m gpu -q -p no:cacheprovider -x -k "not dropout_matches_oracle and not optimizer_update" > gpurun_out/pytest_r05_call6.log 2>&1
echo "pytest rc=$?"; grep -E " passed| failed" gpurun_out/pytest_r05_call6.log | tail -2; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r05_call6.log | head
bash scripts/ab_bench.sh build_ab/libt2v_r05_before.so 2 2>&1 | grep -E "^(old|new)"
cd /tmp && export TMPDIR=/tmp
for variant in "--no-text-encoder" "--no-graph" "--no-text-encoder --no-graph"; do
  T2V_GRAPH_PIPELINE=0 timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_try -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-default-mode --no-host-timing $variant > /tmp/pmc_try.log 2>&1
  echo "pmc attempt [$variant] rc=$? $(grep -c SIGSEGV /tmp/pmc_try.log) segv; $(grep -c '^{' /tmp/pmc_try.log) json lines"
  rm -rf /tmp/pmc_try
done
