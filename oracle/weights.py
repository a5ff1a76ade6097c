"""Seeded synthetic-weight factory shared by the oracle-side tests and the bench CPU baseline.

No checkpoints exist offline: weights are torch default inits under a fixed seed,
except `TemporalConvLayer.conv4` which upstream zero-initialises — that would hide the
whole Conv3d path, so it is drawn N(0, (3C)^-1/2) — and, on request, LoRA `up` factors N(0, 1/r).
"""
import torch
from torch import nn


def randomize_temporal_conv4(model, seed=7):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if m.__class__.__name__ == "TemporalConvLayer":
            conv = m.conv4[-1]
            conv = getattr(conv, "conv", conv)  # LoRA-wrapped
            c = conv.weight.shape[1]
            with torch.no_grad():
                conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (3 * c) ** -0.5)
                conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.02)
    return model


def randomize_lora_up(model, seed=11, scale=1.0):
    """LoRA `up` factors ~ N(0, (scale/r)^2) (the reference initialises them to zero, utils/lora.py:55, which would
    hide the whole LoRA branch from a parity test); scale=0 restores the reference init."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if hasattr(m, "lora_up") and hasattr(m, "lora_down"):
            with torch.no_grad():
                w = m.lora_up.weight
                w.copy_(torch.randn(w.shape, generator=g) * (scale / m.r))
    return model


def make_model(ctor, seed=0, **cfg):
    torch.manual_seed(seed)
    m = ctor(**cfg)
    randomize_temporal_conv4(m, seed + 7)
    return m


def synthetic_batch(frames, height, width, seed=1234, text_dim=1024, latent_down=8, batch=1):
    """SURVEY.md §8d synthetic inputs, all drawn on the host."""
    g = torch.Generator().manual_seed(seed)
    h, w = height // latent_down, width // latent_down
    return dict(
        pixel_values=torch.rand(batch, frames, 3, height, width, generator=g) * 2 - 1,
        encoder_hidden_states=torch.randn(batch, 77, text_dim, generator=g),
        vae_eps=torch.randn(batch * frames, 4, h, w, generator=torch.Generator().manual_seed(seed + 1)),
        noise=torch.randn(batch, 4, frames, h, w, generator=torch.Generator().manual_seed(seed + 2)),
        timesteps=torch.randint(0, 1000, (batch,), generator=torch.Generator().manual_seed(seed + 3)),
    )
