"""Oracle fixtures for the default-train-mode LoRA step on the shipped grids (BASELINE.json configs[3..4]) at reduced width:
    python tests/golden/make_grid_fixture.py            # ~6 minutes of host time, writes tests/golden/oracle_grid_*.pt
The CPU fp32 oracle (oracle/unet3d.py + oracle/lora.py) runs ONE two-pass train step with the reference's default dropout —
LoRA wrappers 0.1 (utils/lora.py:35,89), TemporalConvLayer 0.1 (models/unet_3d_blocks.py:312) — every mask restated element by
element from the native protocol (oracle/dropout.py: base seed, host step 0, device epoch 2 = the first step of a fresh trainer).
Stored: the loss, EVERY LoRA-factor gradient (bf16 — 0.4 % per element against bars of 3 - 25 %; 2 - 4 MB per grid) and a checksum of the seeded weights.  The GPU test
(tests/test_lora_grads_gpu.py::test_default_train_mode_lora_on_the_shipped_grids) rebuilds the same weights, runs the native step
and compares — the oracle itself needs 2.5 - 3.5 minutes per grid, too long for the driver's GPU-test step."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import parity_utils as pu  # noqa: E402

GRIDS = [(8, 40, 72, 16), (4, 48, 128, 32)]          # (frames, latent h, latent w, LoRA rank)
SCALE = 0.05                                         # lora_up ~ N(0, (0.05 / sqrt(r))^2): live branches


def grid_path(frames, h, w, r):
    return os.path.join(HERE, f"oracle_grid_{frames}x{h}x{w}_r{r}_drop.pt")


def grid_batch(frames, h, w, r):
    from oracle.weights import synthetic_batch
    return synthetic_batch(frames, 8 * h, 8 * w, seed=777 + r, text_dim=64)


def main():
    from oracle import dropout as odrop
    for frames, h, w, r in GRIDS:
        ounet, ovae, _ = pu.build_oracle(False, r, SCALE)
        pu.enable_reference_dropout(ounet)
        ctx = odrop.install_protocol(ounet, pu.DROPOUT_BASE_SEED, step=0, epoch=2, batch=1, frames=frames, passes=2)
        checksum = pu.weight_checksum(ounet, ovae)
        loss, grads = pu.oracle_loss_and_grads(ounet, ovae, grid_batch(frames, h, w, r))
        assert ctx["k"] == 1
        gn = float(torch.cat([g.flatten().double() for g in grads.values()]).norm())
        torch.save({"loss": loss, "grads": {n: g.to(torch.bfloat16) for n, g in grads.items()}, "grad_norm": gn, "checksum": checksum,
                    "grid": (frames, h, w, r), "scale": SCALE, "dropout": True}, grid_path(frames, h, w, r))
        print(f"grid {frames}x{h}x{w} r={r}: loss {loss:.6f} |g| {gn:.4e} -> {grid_path(frames, h, w, r)}", flush=True)


if __name__ == "__main__":
    main()
