#!/bin/bash
# Steady-state kernel statistics of the bench: rocprofv3 kernel trace, then per-kernel stats over the last 4 graph replays.
# usage (on the GPU box): bash scripts/profile_bench.sh <tag>   -> gpurun_out/<tag>_window.txt
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace -d $out -o bench -- python $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-default-mode --no-host-timing > $out.log 2>&1
ms=$(grep '^{"metric"' $out.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
db=$(find $out -name '*.db' | head -1)
python $root/scripts/rocpd_window.py $db $(python -c "print(4*$ms)") 4 90 > $root/gpurun_out/${tag}_window.txt 2>&1
python $root/scripts/rocpd_gaps.py $db $(python -c "print(4*$ms)") 4 > $root/gpurun_out/${tag}_gaps.txt 2>&1
rm -rf $out
head -30 $root/gpurun_out/${tag}_window.txt | cut -c1-150
