#!/usr/bin/env python
"""Generate the full-size oracle fixtures `oracle_step_<config>_s<scale>.pt` (run in the build container, CPU only):

    python tests/golden/make_oracle_step.py --config c1 --scales 0,0.02,0.2
    python tests/golden/make_oracle_step.py --config c2 --scales 0.02
    python tests/golden/make_oracle_step.py --config c3            (full finetune: gradients of all 1.41 B UNet parameters, C1 clip)
    python tests/golden/make_oracle_step.py --config c3full        (the same on configs[2]'s own clip: 16 frames @256x256)
    python tests/golden/make_oracle_step.py --config c1 --scales 0.02 --dropout   (default train mode, restated masks)
    python tests/golden/make_oracle_step.py --config c2 --scales 0.02 --dropout   (the same at the benchmark configuration)

For ModelScope-1.7B shapes with host-seeded weights/inputs (tests/parity_utils.py) the CPU fp32 oracle evaluates the
eps-MSE of train.py:793-834 and its gradients w.r.t. all 1148 LoRA factors.  Recorded per fixture:
  loss, weight checksum, per-tensor gradient norms, 4 seeded +-1 random projections of EVERY gradient tensor (a complete
  sketch: a mis-laid-out tensor decorrelates its projections), and the exact values (first 8192 elements) of a sample of
  tensors covering every layer kind.
The GPU tests (tests/test_lora_grads_gpu.py) rebuild the same weights, verify the checksum, and compare the native path.
"""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

KINDS = ("downsamplers", "upsamplers", "conv_shortcut", "time_emb_proj", "temp_convs", "attn1.to_q", "attn1.to_k",
         "attn1.to_v", "attn2.to_q", "attn2.to_k", "attn2.to_v", "to_out", "ff.net.0", "ff.net.2", "proj_in", "proj_out",
         "resnets.0.conv1", "resnets.0.conv2", "transformer_in", "time_embedding", "conv_in", "conv_out", "mid_block")
NPROJ = 4


def projection_signs(name, numel, k=NPROJ):
    """k seeded +-1 vectors for tensor `name` (seed = hash of the name: identical on every machine)."""
    seed = int.from_bytes(hashlib.sha256(name.encode()).digest()[:6], "little")
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 2, (k, numel), generator=g, dtype=torch.int8).float() * 2 - 1


def sketch(name, t):
    v = t.detach().double().flatten()
    return (projection_signs(name, v.numel()).double() @ v).float()


def sample_names(names):
    names = sorted(names)
    pick = set(names[::24])
    for kind in KINDS:
        for role in ("lora_up", "lora_down"):
            for n in names:
                if kind in n and role in n:
                    pick.add(n)
                    break
    return sorted(pick)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c1")
    ap.add_argument("--scales", default="0,0.02,0.2")
    ap.add_argument("--dropout", action="store_true",
                    help="the reference's DEFAULT train mode: LoRA dropout 0.1 + TemporalConvLayer dropout 0.1, with the masks of "
                         "the native protocol restated on the CPU (oracle/dropout.py; first step of a fresh trainer)")
    args = ap.parse_args()
    import parity_utils as pu
    frames, H, W, r = pu.CONFIGS[args.config]
    from oracle.weights import synthetic_batch
    if args.config in ("c3", "c3full"):            # full finetune: one fixture (no LoRA factors to scale)
        args.scales = "0"
    for scale in [float(s) for s in args.scales.split(",")]:
        t0 = time.time()
        if args.config in ("c3", "c3full"):
            unet, vae = pu.build_oracle_full_finetune(True)
            n_wrapped = 0
        else:
            unet, vae, n_wrapped = pu.build_oracle(True, r, scale)
        batch = synthetic_batch(frames, H, W, seed=1234)
        if args.dropout:
            from oracle import dropout as odrop
            assert args.config in ("c1", "c2"), "the two passes draw different masks: both are evaluated (C2: ~25 min on 8 cores)"
            pu.enable_reference_dropout(unet)
            # first _fwd_bwd of a fresh trainer: device epoch = (rank << 32) + 2, host step 0 (tests/test_lora_grads_gpu.py)
            ctx = odrop.install_protocol(unet, pu.DROPOUT_BASE_SEED, step=0, epoch=2, batch=1, frames=frames, passes=2)
        loss, grads = pu.oracle_loss_and_grads(unet, vae, batch, single_pass_doubled=(args.config in ("c2", "c3full") and not args.dropout),
                                               sequential_passes=(args.config == "c2" and args.dropout))
        if args.dropout:
            assert ctx["k"] == 1, "both passes must have run through the protocol"
        total = sum(float(g.double().pow(2).sum()) for g in grads.values()) ** 0.5
        fx = dict(dropout=bool(args.dropout), config=args.config, frames=frames, height=H, width=W, rank=r, lora_up_scale=scale, seed=0, batch_seed=1234,
                  loss=loss, n_wrapped=n_wrapped, checksum=pu.weight_checksum(unet, vae), grad_norm=total,
                  grad_norms={n: float(g.double().norm()) for n, g in grads.items()},
                  sketches={n: sketch(n, g) for n, g in grads.items()},
                  samples={n: grads[n].flatten()[:8192].clone() for n in sample_names(grads)},
                  torch_version=torch.__version__)
        path = pu.fixture_path(args.config, scale, dropout=args.dropout)
        torch.save(fx, path)
        print(f"{path}: loss {loss:.6f} |g| {total:.4e} tensors {len(grads)} samples {len(fx['samples'])} "
              f"({os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
