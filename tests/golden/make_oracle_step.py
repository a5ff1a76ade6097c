#!/usr/bin/env python
"""Generate the full-size oracle fixtures `oracle_step_<config>_s<scale>.pt` (run in the build container, CPU only):

    python tests/golden/make_oracle_step.py --config c1 --scales 0,0.02,0.2
    python tests/golden/make_oracle_step.py --config c2 --scales 0.02
    python tests/golden/make_oracle_step.py --config c3            (full finetune: gradients of all 1.41 B UNet parameters, C1 clip)
    python tests/golden/make_oracle_step.py --config c3full        (the same on configs[2]'s own clip: 16 frames @256x256)
    python tests/golden/make_oracle_step.py --config c1 --scales 0.02 --dropout   (default train mode, restated masks)
    python tests/golden/make_oracle_step.py --config c2 --scales 0.02 --dropout   (the same at the benchmark configuration)
    ... --floor        (C1, no dropout) also runs the oracle under torch.autocast(cpu, bf16) — the reference's own recipe,
                       train.py:848-852 — and records ITS per-tensor error for the tensors stored in full

For ModelScope-1.7B shapes with host-seeded weights/inputs (tests/parity_utils.py) the CPU fp32 oracle evaluates the
eps-MSE of train.py:793-834 and its gradients w.r.t. all 1148 LoRA factors.  Recorded per fixture:
  loss, weight checksum, per-tensor gradient norms, 4 seeded +-1 random projections of EVERY gradient tensor (a complete
  sketch: a mis-laid-out tensor decorrelates its projections), the exact values (first 8192 elements) of a sample of
  tensors covering every layer kind, and (round 5) `big`: the COMPLETE fp32 gradient of every tensor holding >= 1 % of the
  gradient norm — the per-tensor gate compares these element by element (a 4-projection estimate of one tensor's error is a
  chi^2_4 variable: it reads 0.4x .. 1.6x the true value, which is what tripped the round-4 gate) — with `big_floor`, the
  error of the bf16-autocast oracle on the same tensors where `--floor` was given.
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


def sketch_big(name, t):
    """Count-sketch of a large tensor: one seeded +-1 sign per element, elements folded into NPROJ_BIG buckets
    (y_j = sum over i = j mod NPROJ_BIG of d_i v_i).  E sum_j (y_j - y'_j)^2 = ||v - v'||^2 with the spread of a chi^2_64 variable
    (+-18 % on the square, +-9 % on the error itself), at one random draw per element (a 14.7 M-element weight: 0.2 s)."""
    v = t.detach().double().flatten()
    seed = int.from_bytes(hashlib.sha256(("big:" + name).encode()).digest()[:6], "little")
    g = torch.Generator().manual_seed(seed)
    d = torch.randint(0, 2, (v.numel(),), generator=g, dtype=torch.int8).double() * 2 - 1
    w = d * v
    pad = (-w.numel()) % NPROJ_BIG
    if pad:
        w = torch.cat([w, w.new_zeros(pad)])
    return w.view(-1, NPROJ_BIG).sum(0)


BIG_SHARE = 1e-2        # tensors holding >= 1 % of the gradient norm are stored in full ...
BIG_FULL_NUMEL = 65536  # ... up to this size; larger ones (rank-16 down factors of 3x3 convs, full-finetune weights) as
NPROJ_BIG = 64          # a 64-bucket count-sketch: the error estimate of ONE tensor is then good to +-9 % (chi^2_64) instead of +-2x


def autocast_floor_of(unet, vae, batch, big):
    """Per-tensor relative error of the SAME oracle under torch.autocast(cpu, bfloat16) (the reference's mixed-precision
    recipe, train.py:848-852; fp32 latents for both arms as in scripts/autocast_floor.py) on the tensors of `big`."""
    from oracle.fastconv import fast_temporal_conv3d
    from oracle.train_step import finetune_unet_loss
    from oracle.vae import tensor_to_vae_latent
    for p in unet.parameters():
        p.grad = None
    with torch.no_grad():
        latents = tensor_to_vae_latent(batch["pixel_values"], vae, batch["vae_eps"])
    with fast_temporal_conv3d():
        with torch.autocast("cpu", dtype=torch.bfloat16):
            loss, _ = finetune_unet_loss(unet, vae, batch, cached_latents=latents)
        loss.backward()
    g16 = {n: p.grad.detach().double() for n, p in unet.named_parameters() if p.requires_grad and n in big}
    return {n: float((g16[n] - v.double()).norm() / v.double().norm()) for n, v in big.items()}


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
    ap.add_argument("--floor", action="store_true",
                    help="also run the oracle under torch.autocast(cpu, bfloat16) and record its per-tensor error on the `big` tensors")
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
        big_names = [n for n, g in grads.items() if float(g.double().pow(2).sum()) >= BIG_SHARE ** 2 * total ** 2]
        big = {n: grads[n].detach().float().clone() for n in big_names if grads[n].numel() <= BIG_FULL_NUMEL}
        big_sketch = {n: sketch_big(n, grads[n]) for n in big_names if grads[n].numel() > BIG_FULL_NUMEL}
        big_floor = {}
        if args.floor:
            assert not args.dropout and args.config == "c1", "the floor arm re-runs the two-pass step: C1, eval_train mode"
            big_floor = autocast_floor_of(unet, vae, batch, {n: grads[n] for n in big_names})
        fx = dict(dropout=bool(args.dropout), config=args.config, frames=frames, height=H, width=W, rank=r, lora_up_scale=scale, seed=0, batch_seed=1234,
                  loss=loss, n_wrapped=n_wrapped, checksum=pu.weight_checksum(unet, vae), grad_norm=total,
                  grad_norms={n: float(g.double().norm()) for n, g in grads.items()},
                  sketches={n: sketch(n, g) for n, g in grads.items()},
                  samples={n: grads[n].flatten()[:8192].clone() for n in sample_names(grads)},
                  big=big, big_sketch=big_sketch, big_floor=big_floor, big_share=BIG_SHARE,
                  torch_version=torch.__version__)
        path = pu.fixture_path(args.config, scale, dropout=args.dropout)
        torch.save(fx, path + ".tmp")
        os.replace(path + ".tmp", path)          # (atomic: a gpurun snapshot taken meanwhile never sees half a file)
        print(f"{path}: loss {loss:.6f} |g| {total:.4e} tensors {len(grads)} samples {len(fx['samples'])} big {len(big)}+{len(big_sketch)} "
              f"floor max {max(big_floor.values()) if big_floor else float('nan'):.3f} "
              f"({os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
