#!/bin/bash
# HBM-side traffic of EVERY kernel of the C2 train step (the tiles the step actually runs, shipped tile table):
# rocprofv3 --kernel-trace --pmc FETCH_SIZE, then --pmc WRITE_SIZE (separate passes; no other trace domains), 2 graph-replayed
# steps each.  Per-kernel totals -> gpurun_out/<tag>_{FETCH_SIZE,WRITE_SIZE}.txt (scripts/rocpd_pmc.py); summarised into
# profiles/ by scripts/pmc_summary.py.   usage (GPU box): bash scripts/pmc_step.sh <tag>
tag=${1:-pmc_step}
# counter collection serialises dispatches: a two-stream (pipelined) replay then dead-locks on its cross-stream events
# (r04: 1666 incomplete dispatches after 8 minutes), and in round 5 the tool crashed at start-up (SIGSEGV in its own thread, two
# boxes) whenever the CLIP tower was CAPTURED under it — graph replay without the text encoder and the eager step with it both
# collect fine (scripts/r05_call6.sh).  The counter passes therefore run the EAGER step (--no-graph): the same kernels on the same
# tiles, one dispatch after the other, which is what the per-kernel byte counts need.
export T2V_GRAPH_PIPELINE=0
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/${tag}_$c
  timeout 270 rocprofv3 --kernel-trace --pmc $c -d $out -o pmc -- python $root/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-default-mode --no-host-timing > $out.log 2>&1
  ms=$(grep '^{"metric"' $out.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  db=$(find $out -name '*.db' | head -1)
  # the last 3 steps of the trace = the timed graph replays (window = 3 x the step time the profiled run itself measured)
  python $root/scripts/rocpd_pmc.py $db 160 $(python -c "print(3*$ms)") > $root/gpurun_out/${tag}_$c.txt 2>&1
  echo "steps_in_window 3 ms_per_step_under_pmc $ms" >> $root/gpurun_out/${tag}_$c.txt
  rm -rf $out
done
head -12 $root/gpurun_out/${tag}_FETCH_SIZE.txt | cut -c1-170
