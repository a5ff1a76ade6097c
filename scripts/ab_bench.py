"""A/B the train step under two environment settings inside ONE gpurun call (box-to-box spread of the same build is ±5 %, so
numbers from different calls cannot be compared).  Runs bench.py alternately A, B, A, B and prints ms/step of every run.

    python scripts/ab_bench.py "T2V_WGRAD_STREAM=1" "T2V_WGRAD_STREAM=0" [--steps 5] [--rounds 2] [--config c2]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(envs, steps, config):
    env = dict(os.environ)
    for kv in envs.split():
        k, _, v = kv.partition("=")
        env[k] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1", "--config", config,
                          "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        raise SystemExit(f"bench failed under [{envs}]:\n{out.stderr[-2000:]}")
    return json.loads(lines[-1])["ms_per_step"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("env_a")
    ap.add_argument("env_b")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--config", default="c2")
    a = ap.parse_args()
    res = {"A": [], "B": []}
    for _ in range(a.rounds):
        res["A"].append(run(a.env_a, a.steps, a.config))
        res["B"].append(run(a.env_b, a.steps, a.config))
    for k, e in (("A", a.env_a), ("B", a.env_b)):
        v = res[k]
        print(f"{k} [{e}]: ms/step {v}  best {min(v):.2f}")
    print(f"B/A (best): {min(res['B']) / min(res['A']):.4f}")


if __name__ == "__main__":
    main()
