#!/bin/bash
# Kernel statistics of any bench configuration (steady-state graph replays): bash scripts/profile_config.sh <tag> <bench args...>
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out -o bench -- python $root/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-default-mode --no-host-timing > $out.log 2>&1
ms=$(grep '^{"metric"' $out.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
db=$(find $out -name '*.db' | head -1)
python $root/scripts/rocpd_window.py $db $(python -c "print(3*$ms)") 3 60 > $root/gpurun_out/${tag}_window.txt 2>&1
rm -rf $out
head -40 $root/gpurun_out/${tag}_window.txt | cut -c1-150
