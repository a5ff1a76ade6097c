"""CPU fp32 restatement of one optimisation step (ORACLE — test infrastructure).

Follows `train.py:720-836` (`finetune_unet`) and `train.py:848-879` (backward, clip, AdamW):
  latents  = tensor_to_vae_latent(pixel_values)           train.py:742  (339-347)
  noise    ~ N(0,1), t ~ U{0..999}                         train.py:751-756   (host-injected here)
  noisy    = add_noise(latents, noise, t)                  train.py:760
  two UNet passes, loss = mse0 + mse1                      train.py:814-834
  backward; clip_grad_norm_(unet.parameters(), 1.0); AdamW train.py:861-879
The text encoder output is an input (`encoder_hidden_states`), SURVEY.md §8(f) row 2.
"""
import torch
import torch.nn.functional as F

from . import scheduler
from .vae import tensor_to_vae_latent


def finetune_unet_loss(unet, vae, batch, acp=None, cached_latents=None):
    if cached_latents is None:
        with torch.no_grad():
            latents = tensor_to_vae_latent(batch["pixel_values"], vae, batch["vae_eps"])
    else:
        latents = cached_latents
    noise, timesteps = batch["noise"], batch["timesteps"]
    noisy = scheduler.add_noise(latents, noise, timesteps, acp)
    target = noise  # epsilon prediction (train.py:793-794)
    ehs = batch["encoder_hidden_states"]
    video_length = latents.shape[2]
    losses = []
    for i in range(2):
        pred = unet(noisy, timesteps, encoder_hidden_states=ehs).sample
        losses.append(F.mse_loss(pred.float(), target.float(), reduction="mean"))
        if video_length == 1 and i == 0:
            break
    loss = losses[0] if len(losses) == 1 else losses[0] + losses[1]
    return loss, latents


def train_step(unet, vae, batch, optimizer, max_grad_norm=1.0, acp=None):
    loss, latents = finetune_unet_loss(unet, vae, batch, acp)
    loss.backward()
    if max_grad_norm > 0:
        torch.nn.utils.clip_grad_norm_(list(unet.parameters()), max_grad_norm)
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return loss.detach(), latents
