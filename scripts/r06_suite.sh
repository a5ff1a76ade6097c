#!/bin/bash
# Round 6: the WHOLE GPU suite without -x, log + parity rows into gpurun_out/.
#   gpurun --timeout 1500 -- 'bash scripts/r06_suite.sh first'
tag=${1:-run}
mkdir -p gpurun_out
rm -f gpurun_out/parity_r06.jsonl
t0=$(date +%s)
timeout 1400 python -m pytest tests -m gpu -q -rA --durations=25 -p no:cacheprovider > gpurun_out/pytest_r06_${tag}.log 2>&1
rc=$?
echo "pytest rc=$rc wall=$(( $(date +%s) - t0 )) s" | tee -a gpurun_out/pytest_r06_${tag}.log
grep -E "passed|failed" gpurun_out/pytest_r06_${tag}.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r06_${tag}.log | head -40
exit $rc
