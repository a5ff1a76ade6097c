#!/usr/bin/env python
"""Noise floor of the REFERENCE's own mixed-precision recipe, measured on the CPU oracle (test infrastructure).

The reference trains under `accelerator.autocast()` (train.py:848-852): fp32 master weights, bf16 (or fp16) matmul /
conv operands and activations.  This script runs the CPU fp32 oracle and the SAME oracle under
`torch.autocast("cpu", torch.bfloat16)` on identical seeded inputs and reports, per LoRA `up` amplitude
(oracle.weights.randomize_lora_up scale):
    loss_rel      |L_bf16 - L_fp32| / |L_fp32|                       (eps-MSE, two-pass sum)
    grad_rel      ||g_bf16 - g_fp32|| / ||g_fp32||  over all LoRA factor gradients (one flat vector)
    grad_rel_max  worst per-tensor relative error among tensors holding >= 1e-3 of the gradient norm
These are the numbers the GPU parity tolerances are anchored to (tests/test_parity_floor.py, tests/test_unet_gpu.py).

    python scripts/autocast_floor.py [--full-c1] [--scales 0,0.02,0.2,1] [--out tests/golden/autocast_floor.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

SMALL = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, attention_head_dim=64)
VAE_SMALL = dict(block_out_channels=(32, 64, 64, 64))


def build(full, r, scale):
    from oracle.lora import inject_trainable_lora_extended
    from oracle.unet3d import UNet3DConditionModel
    from oracle.vae import AutoencoderKLEncoder
    from oracle.weights import randomize_lora_up, randomize_temporal_conv4
    torch.manual_seed(0)
    unet = UNet3DConditionModel(**({} if full else SMALL))
    randomize_temporal_conv4(unet)
    vae = AutoencoderKLEncoder(**({} if full else VAE_SMALL)).eval()
    unet.requires_grad_(False)
    vae.requires_grad_(False)
    inject_trainable_lora_extended(unet, {"UNet3DConditionModel"}, r=r)
    randomize_lora_up(unet, scale=scale)
    for m in unet.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    unet.train()
    return unet, vae


def run(unet, vae, batch, latents, bf16):
    from oracle.fastconv import fast_temporal_conv3d
    from oracle.train_step import finetune_unet_loss
    for p in unet.parameters():
        p.grad = None
    with fast_temporal_conv3d():
        if bf16:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                loss, _ = finetune_unet_loss(unet, vae, batch, cached_latents=latents)
        else:
            loss, _ = finetune_unet_loss(unet, vae, batch, cached_latents=latents)
        loss.backward()
    grads = {n: p.grad.detach().double().clone() for n, p in unet.named_parameters() if p.requires_grad}
    return float(loss.detach()), grads


def compare(g_ref, g_dut):
    num = sum(float((g_dut[n] - g_ref[n]).pow(2).sum()) for n in g_ref)
    den = sum(float(g_ref[n].pow(2).sum()) for n in g_ref)
    worst = 0.0
    for n in g_ref:
        nn_ = float(g_ref[n].pow(2).sum())
        if nn_ >= 1e-6 * den and nn_ > 0:            # tensors holding >= 1e-3 of the gradient NORM
            worst = max(worst, (float((g_dut[n] - g_ref[n]).pow(2).sum()) / nn_) ** 0.5)
    return (num / max(den, 1e-300)) ** 0.5, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-c1", action="store_true", help="ModelScope-1.7B shapes at config C1 (minutes per scale)")
    ap.add_argument("--scales", default="0,0.02,0.2,1")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from oracle.vae import tensor_to_vae_latent
    from oracle.weights import synthetic_batch
    full = args.full_c1
    rows = []
    for scale in [float(s) for s in args.scales.split(",")]:
        t0 = time.time()
        unet, vae = build(full, 4, scale)
        batch = synthetic_batch(8, 128, 128, seed=1234) if full else synthetic_batch(4, 64, 64, seed=100, text_dim=64)
        with torch.no_grad():
            latents = tensor_to_vae_latent(batch["pixel_values"], vae, batch["vae_eps"])   # fp32 latents for both arms
        l32, g32 = run(unet, vae, batch, latents, False)
        l16, g16 = run(unet, vae, batch, latents, True)
        grel, gmax = compare(g32, g16)
        row = dict(config="c1-full" if full else "toy", lora_up_scale=scale, loss_fp32=l32, loss_bf16_autocast=l16,
                   loss_rel=abs(l16 - l32) / abs(l32), grad_rel=grel, grad_rel_max=gmax, seconds=round(time.time() - t0, 1))
        print(json.dumps(row), flush=True)
        rows.append(row)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(dict(note="CPU oracle fp32 vs the same oracle under torch.autocast(cpu, bfloat16): the noise floor "
                                "of the reference's own mixed-precision recipe (scripts/autocast_floor.py)", rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
