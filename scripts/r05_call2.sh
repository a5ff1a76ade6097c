#!/bin/bash
# Round 5, GPU call 2: the two tests fixed after the first full run + the new ones, the regenerated full-size fixtures, then the
# dispatch-policy A/B of the dropped LoRA branch (whole step, same process) and its per-shape probe.
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "cached_latents or gradient_checkpointing or state_round_trip or zz_fullsize" -rA > gpurun_out/pytest_r05_call2.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_r05_call2.log | tail -2; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r05_call2.log | head
timeout 600 python scripts/policy_ab.py --steps 30 --rounds 2 > gpurun_out/r05_policy_ab.txt 2> gpurun_out/r05_policy_ab.err
echo "policy_ab rc=$?"; tail -12 gpurun_out/r05_policy_ab.txt; tail -3 gpurun_out/r05_policy_ab.err
timeout 300 python scripts/branch_probe.py > gpurun_out/r05_branch_probe.txt 2> gpurun_out/r05_branch_probe.err
echo "branch_probe rc=$?"; cat gpurun_out/r05_branch_probe.txt; tail -3 gpurun_out/r05_branch_probe.err
