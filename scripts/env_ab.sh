#!/bin/bash
# Step time of the default bench under HIP-runtime environment settings, one box, alternated twice:
#   gpurun -- 'bash scripts/env_ab.sh "" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" ...'
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for round in 1 2; do
  for e in "$@"; do
    ms=$(env $e timeout 200 python bench.py --steps ${STEPS:-30} --warmup 2 --no-cpu-baseline --no-roofline --no-default-mode --no-host-timing 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $round [$e] ms/step $ms"
  done
done
