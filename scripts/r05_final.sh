#!/bin/bash
# Round 5 final measurement set on ONE box:  bash scripts/r05_final.sh <tag> [nosuite] [nopmc]
#   1. the WHOLE GPU suite without -x (log -> gpurun_out/pytest_r05_<tag>.log, parity rows -> gpurun_out/parity_r05.jsonl)
#   2. python bench.py (the driver's command: default train mode at C2, roofline + cpu_baseline) -> gpurun_out/r05_bench_default_<tag>.json
#   3. rocprofv3 kernel trace of the replayed step -> gpurun_out/<tag>_window.txt
#   4. HBM-side traffic: two --pmc passes over the EAGER step (scripts/pmc_step.sh says why)
tag=${1:-final}
mkdir -p gpurun_out
if [ "${2:-}" != "nosuite" ]; then
  rm -f gpurun_out/parity_r05.jsonl
  t0=$(date +%s)
  timeout 1500 python -m pytest tests -m gpu -q -rA --durations=15 -p no:cacheprovider > gpurun_out/pytest_r05_${tag}.log 2>&1
  echo "pytest rc=$? wall=$(( $(date +%s) - t0 )) s" | tee -a gpurun_out/pytest_r05_${tag}.log
  grep -E " passed| failed" gpurun_out/pytest_r05_${tag}.log | tail -2
  grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r05_${tag}.log | head -20
fi
timeout 600 python bench.py > gpurun_out/r05_bench_default_${tag}.json 2> gpurun_out/r05_bench_default_${tag}.err
echo "bench rc=$?"; grep '^{' gpurun_out/r05_bench_default_${tag}.json | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('ms/step', d['ms_per_step'], 'videos/s', d['value'], 'pipelined', d['config'].get('graph_pipeline'), 'eval_train ms', d['config'].get('eval_train_ms_per_step'),
      'roofline frac', (d.get('roofline') or {}).get('frac'), 'gemm ms', (d.get('roofline') or {}).get('kernel_ms_per_step'), 'eps rel', d['config'].get('eps_mse_rel_err'))"
tail -2 gpurun_out/r05_bench_default_${tag}.err
bash scripts/profile_bench.sh ${tag} > /dev/null 2>&1; head -8 gpurun_out/${tag}_window.txt | cut -c1-140
if [ "${3:-}" != "nopmc" ]; then
  bash scripts/pmc_step.sh ${tag}_pmc 2>&1 | tail -14
fi
