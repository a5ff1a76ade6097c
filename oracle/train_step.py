"""CPU fp32 restatement of one optimisation step (ORACLE — test infrastructure).

Follows `train.py:720-836` (`finetune_unet`) and `train.py:848-879` (backward, clip, AdamW):
  latents  = tensor_to_vae_latent(pixel_values)           train.py:742  (339-347)
  noise    ~ N(0,1), t ~ U{0..999}                         train.py:751-756   (host-injected here)
  noisy    = add_noise(latents, noise, t)                  train.py:760
  two UNet passes, loss = mse0 + mse1                      train.py:814-834
  backward; clip_grad_norm_(unet.parameters(), 1.0); AdamW train.py:861-879
The text encoder output is an input (`encoder_hidden_states`) unless a `text_encoder` is passed; with a TRAINABLE text
encoder the reference's two passes differ (train.py:805-828): pass 0 sees the whole clip with DETACHED text states, pass 1
sees frame 1 only with the trainable states ("train text information only on the spatial layers").
"""
import torch
import torch.nn.functional as F

from . import scheduler
from .vae import tensor_to_vae_latent


def finetune_unet_loss(unet, vae, batch, acp=None, cached_latents=None, text_encoder=None, text_trainable=False):
    if cached_latents is None:
        with torch.no_grad():
            latents = tensor_to_vae_latent(batch["pixel_values"], vae, batch["vae_eps"])
    else:
        latents = cached_latents
    noise, timesteps = batch["noise"], batch["timesteps"]
    noisy = scheduler.add_noise(latents, noise, timesteps, acp)
    target = noise  # epsilon prediction (train.py:793-794)
    if text_encoder is not None:
        ids = batch["prompt_ids"]
        ehs = text_encoder(ids[0] if ids.dim() > 2 else ids)[0]         # train.py:784-790
    else:
        ehs = batch["encoder_hidden_states"]
    video_length = latents.shape[2]
    should_truncate = video_length > 1 and text_trainable               # train.py:807
    detached, trainable = ehs.clone().detach(), ehs.clone()             # train.py:811-812
    losses = []
    for i in range(2):
        should_detach = noisy.shape[2] > 1 and i == 0                    # train.py:816
        if should_truncate and i == 1:                                   # train.py:818-820
            noisy = noisy[:, :, 1, :, :].unsqueeze(2)
            target = target[:, :, 1, :, :].unsqueeze(2)
        ehs = detached if should_detach else trainable
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


def sample_noise(latents, noise_strength, use_offset_noise=False, generator=None):
    """train.py:349-358: eps ~ N(0,1) shaped like the latents; with offset noise, plus `noise_strength` x one N(0,1) draw per
    (b, c, f), broadcast over the grid.  Draw ORDER matters for parity with a shared RNG stream: the full-size noise first,
    then the (b,c,f,1,1) offsets."""
    b, c, f = latents.shape[:3]
    noise = torch.randn(latents.shape, generator=generator, device=latents.device, dtype=latents.dtype)
    if use_offset_noise:
        noise = noise + noise_strength * torch.randn(b, c, f, 1, 1, generator=generator, device=latents.device, dtype=latents.dtype)
    return noise
