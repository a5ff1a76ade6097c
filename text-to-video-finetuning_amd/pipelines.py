"""Latent-space sampling loop over the drop-in UNet: forward-only reuse of the training kernels with the
classifier-free-guidance batch doubling of `TextToVideoSDPipeline` (train.py:918-943) / `inference.py:153-267`.

Scope: prompt embeddings in, denoised latents `(B,4,F,h,w)` out.  Decoding the latents to frames needs the VAE decoder, which
is not built yet (the train step only encodes); `decode=True` raises instead of silently returning something else."""
import torch

from .schedulers import DPMSolverMultistepScheduler


class TextToVideoSampler:
    def __init__(self, unet, scheduler=None, vae=None):
        self.unet, self.vae = unet, vae
        self.scheduler = scheduler or DPMSolverMultistepScheduler()

    @torch.no_grad()
    def __call__(self, prompt_embeds, negative_prompt_embeds=None, num_frames=16, height=256, width=256,
                 num_inference_steps=25, guidance_scale=9.0, generator=None, latents=None, decode=False):
        if decode:
            raise NotImplementedError("t2v_amd: the VAE decoder is not built (latents only)")
        b = prompt_embeds.shape[0]
        dev = prompt_embeds.device
        cfg = guidance_scale > 1.0 and negative_prompt_embeds is not None
        if latents is None:
            shape = (b, self.unet.config.in_channels, num_frames, height // 8, width // 8)
            latents = torch.randn(shape, generator=generator, device=generator.device if generator is not None else "cpu").to(dev)
        ehs = torch.cat([negative_prompt_embeds, prompt_embeds], 0) if cfg else prompt_embeds
        for t in self.scheduler.set_timesteps(num_inference_steps):
            x = torch.cat([latents, latents], 0) if cfg else latents            # CFG: unconditional + conditional in one forward
            ts = torch.full((x.shape[0],), int(t), dtype=torch.long, device=dev)
            eps = self.unet(x, ts, encoder_hidden_states=ehs).sample
            if cfg:
                e_u, e_c = eps.chunk(2)
                eps = e_u + guidance_scale * (e_c - e_u)
            latents = self.scheduler.step(eps.to(latents.dtype), t, latents)
        return latents
