#!/bin/bash
# Round-4 record set on ONE box: kernel stats of the default-mode bench, PMC traffic, the default bench line (with roofline + CPU
# baseline), the other configurations' lines.   gpurun -- 'bash scripts/r04_measure.sh [quick]'
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
bash scripts/profile_bench.sh r04_bench_c2 > /dev/null 2>&1
bash scripts/pmc_step.sh r04_pmc_step > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r04_pmc_step gpurun_out/r04_pmc_step.json > gpurun_out/r04_pmc_summary.txt 2>&1
# the bench reads the two files above from profiles/: put this run's copies there for the default line
cp gpurun_out/r04_bench_c2_window.txt profiles/r04_bench_c2_kernel_stats.txt
cp gpurun_out/r04_pmc_step.json profiles/r04_pmc_step.json
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
if [ "${1:-}" != "quick" ]; then
  python bench.py --eval-train --no-cpu-baseline --steps 30 > gpurun_out/r04_bench_eval_train.json 2>/dev/null
  python bench.py --config c1 --no-cpu-baseline --steps 30 > gpurun_out/r04_bench_c1.json 2>/dev/null
  python bench.py --config c3 --no-cpu-baseline --steps 10 > gpurun_out/r04_bench_c3.json 2>/dev/null
  python bench.py --config c4 --no-cpu-baseline --steps 10 > gpurun_out/r04_bench_c4.json 2>/dev/null
  python bench.py --config c5 --grad-checkpointing --no-cpu-baseline --no-roofline --steps 5 > gpurun_out/r04_bench_c5_ckpt.json 2>/dev/null
fi
for f in gpurun_out/r04_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1], d["ms_per_step"], d["value"], "roofline", r.get("frac"), "traffic", r.get("traffic"), "eval", d["config"].get("eval_train_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
