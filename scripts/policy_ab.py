#!/usr/bin/env python
"""Same-process A/B of host-side dispatch policies on the benchmark step (config C2, the reference's default train mode): the
model and trainer are built ONCE, every policy sets a few module-level switches of t2v_amd.functional, re-captures the step graph
and times `--steps` replays (HIP events around the whole run).  Policies run round-robin `--rounds` times so that clock drift of
the box shows up as spread inside each row instead of as a difference between rows.

    python scripts/policy_ab.py [--steps 30] [--rounds 2] [--policies name=K:V,K:V ...]
Built-in policy set: where the dropped LoRA branch rides as an epilogue term of the base launch (T2VGemm.lr_mode) and where it runs
as a rank-update pass behind a plain launch, by row count of the layer.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

DEFAULT = [
    ("epi_all(now)", {"_LORA_EPI_MIN_ROWS": 128, "_LORA_EPI_MIN_ROWS_BWD": 128}),
    ("epi_fwd>=16k", {"_LORA_EPI_MIN_ROWS": 16384, "_LORA_EPI_MIN_ROWS_BWD": 128}),
    ("epi_bwd>=16k", {"_LORA_EPI_MIN_ROWS": 128, "_LORA_EPI_MIN_ROWS_BWD": 16384}),
    ("epi_both>=16k", {"_LORA_EPI_MIN_ROWS": 16384, "_LORA_EPI_MIN_ROWS_BWD": 16384}),
    ("epi_both>=4k", {"_LORA_EPI_MIN_ROWS": 4096, "_LORA_EPI_MIN_ROWS_BWD": 4096}),
    ("epi_none", {"_LORA_EPI_MIN_ROWS": 1 << 30, "_LORA_EPI_MIN_ROWS_BWD": 1 << 30}),
    ("now+pipelined", {"_LORA_EPI_MIN_ROWS": 128, "_LORA_EPI_MIN_ROWS_BWD": 128, "pipelined": 1}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--policies", nargs="*", default=None, help="name=KEY:VALUE,KEY:VALUE (integers; keys = attributes of t2v_amd.functional)")
    a = ap.parse_args()
    import t2v_amd  # noqa: F401
    from t2v_amd import functional as F
    from t2v_amd.training import DenoiseTrainer
    pols = DEFAULT
    if a.policies:
        pols = []
        for spec in a.policies:
            name, _, kv = spec.partition("=")
            pols.append((name, {k: int(v) for k, v in (x.split(":") for x in kv.split(",") if x)}))
    dev = torch.device("cuda", 0)
    frames, H, W, r = bench.CONFIGS[a.config]
    unet, vae, trainable = bench.build_models(frames, r, dev, seed=0, dropout=True)
    te = bench.build_text_encoder(dev)
    trainer = DenoiseTrainer(unet, vae, trainable, lr=5e-6, text_encoder=te)
    batch = bench.synthetic_batch(frames, H, W, dev, seed=1234, with_ids=te is not None)
    res = {n: [] for n, _ in pols}
    for rnd in range(a.rounds):
        for name, kv in pols:
            for k, v in kv.items():
                if k == "pipelined":
                    continue
                assert hasattr(F, k), k
                setattr(F, k, v)
            trainer.capture(batch, warmup=1, pipelined=bool(kv.get("pipelined", 0)))
            for _ in range(3):
                trainer.replay_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                loss = trainer.replay_step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            res[name].append(round(ms, 2))
            print(f"round {rnd} {name:16s} {ms:7.2f} ms/step  loss {float(loss):.5f}", flush=True)
            trainer.check_device_flags()
    base = min(res[pols[0][0]])
    print("---- policy: ms/step per round | best | vs first row")
    for name, _ in pols:
        print(f"{name:16s} {res[name]}  best {min(res[name]):.2f}  x{min(res[name]) / base:.4f}")


if __name__ == "__main__":
    main()
