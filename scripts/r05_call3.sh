#!/bin/bash
# Round 5, GPU call 3: new dt kernel + pipelined capture as the default.
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider -rA -k "drop_dt or graph or replay or captured or reload or pipelined or state_round_trip or stable_lora or dropout or test_dp_gpu" > gpurun_out/pytest_r05_call3.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_r05_call3.log | tail -2; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r05_call3.log | head
timeout 400 python scripts/dt_probe.py > gpurun_out/r05_dt_probe.txt 2> gpurun_out/r05_dt_probe.err
echo "dt_probe rc=$?"; cat gpurun_out/r05_dt_probe.txt; tail -3 gpurun_out/r05_dt_probe.err
timeout 400 python scripts/policy_ab.py --steps 30 --rounds 2 --policies "one_graph=pipelined:0" "pipelined=pipelined:1" > gpurun_out/r05_policy_ab3.txt 2> gpurun_out/r05_policy_ab3.err
echo "policy_ab rc=$?"; tail -6 gpurun_out/r05_policy_ab3.txt; tail -3 gpurun_out/r05_policy_ab3.err
