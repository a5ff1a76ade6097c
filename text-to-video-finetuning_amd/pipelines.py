"""Sampling over the drop-in UNet: forward-only reuse of the training kernels with the classifier-free-
guidance batch doubling of `TextToVideoSDPipeline` (train.py:918-943) and the long-video machinery of `inference.py`:

* `diffuse` (inference.py:153-267): every timestep denoises the clip in WINDOWS of `window_size` frames, one UNet call per
  window, with the multistep solver's history kept per frame by the caller (the scheduler object only ever sees one window) and,
  with `rotate`, the clip rolled along the frame axis by a different prime shift per timestep so window borders move;
* `decode` (inference.py:125-140): latents -> frames through the VAE decoder, `vae_batch_size` frames per call.

Prompt embeddings in, denoised latents `(B,4,F,h,w)` — or, with `decode=True`, fp32 frames `(B,3,F,H,W)` in [-1,1] — out."""
import torch

from .schedulers import DPMSolverMultistepScheduler


def primes_up_to(n):
    """The shifts `diffuse` draws from when `rotate` is on (inference.py:143-150): all primes <= n (2 and 3 always included)."""
    out = [p for p in range(2, max(int(n), 3) + 1) if all(p % q for q in range(2, int(p ** 0.5) + 1))]
    return out


class TextToVideoSampler:
    """`graph` (default: on, T2V_SAMPLER_GRAPH=0 turns it off): the UNet call of a denoising step — the same kernels with the same
    shapes at every one of the 25 - 50 timesteps — is captured into a HIP graph on its first use and replayed afterwards; an eager
    no-grad forward of the 1.7B UNet is ~1 700 launches whose ISSUE takes longer than their execution (bench.py
    `sampling_unet_forward_ms`).  The capture is keyed by the call's shapes and is dropped when any UNet parameter has changed
    since (sampling between training steps, train.py:895-958) or the module is in train mode (active dropout draws host-seeded
    masks: eager only)."""

    def __init__(self, unet, scheduler=None, vae=None, graph=None):
        import os
        self.unet, self.vae = unet, vae
        self.scheduler = scheduler or DPMSolverMultistepScheduler()
        self.graph = ((os.environ.get("T2V_SAMPLER_GRAPH", "1") != "0") if graph is None else bool(graph)) and isinstance(unet, torch.nn.Module)
        self._graphs = {}
        self._weights_tag = None

    def _unet_eps(self, xin, ts, ehs):
        return self.unet(xin, ts, encoder_hidden_states=ehs).sample

    def _weights_signature(self):
        # torch's version counters see load_state_dict / a torch optimiser; this library's own optimiser kernels (also inside
        # replayed graphs) are seen through functional.weights_epoch
        from . import functional as F
        return (sum(p._version for p in self.unet.parameters()), sum(1 for _ in self.unet.parameters()), F.weights_epoch[0])

    def _eps_replayed(self, xin, t, ehs):
        """eps of the captured UNet call, or None when this call has to run eagerly (CPU oracle module, train mode)."""
        if not self.graph or not xin.is_cuda or self.unet.training or not hasattr(torch.cuda, "CUDAGraph"):
            return None
        key = (tuple(xin.shape), xin.dtype, tuple(ehs.shape), ehs.dtype, xin.device.index)
        hit = self._graphs.get(key)
        if hit is None:
            st = {"x": xin.clone(), "t": torch.full((xin.shape[0],), int(t), dtype=torch.long, device=xin.device), "ehs": ehs.clone()}
            self._unet_eps(st["x"], st["t"], st["ehs"])          # warm-up: prepared / folded weight copies, workspaces
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["eps"] = self._unet_eps(st["x"], st["t"], st["ehs"])
            hit = (g, st)
            self._graphs[key] = hit
        g, st = hit
        st["x"].copy_(xin)
        st["t"].fill_(int(t))
        st["ehs"].copy_(ehs)
        g.replay()
        return st["eps"]

    def _eps(self, x, t, ehs, cfg, guidance_scale, dev):
        xin = torch.cat([x, x], 0) if cfg else x                               # CFG: unconditional + conditional in one forward
        eps = self._eps_replayed(xin, t, ehs)
        if eps is None:
            ts = torch.full((xin.shape[0],), int(t), dtype=torch.long, device=dev)
            eps = self._unet_eps(xin, ts, ehs)
        if cfg:
            e_u, e_c = eps.chunk(2)
            eps = e_u + guidance_scale * (e_c - e_u)
        else:
            eps = eps.clone()                                                   # (a replay overwrites the captured output)
        return eps.to(x.dtype)

    @torch.no_grad()
    def __call__(self, prompt_embeds, negative_prompt_embeds=None, num_frames=16, height=256, width=256,
                 num_inference_steps=25, guidance_scale=9.0, generator=None, latents=None, decode=False,
                 window_size=None, rotate=False, vae_batch_size=8, init_weight=0.0):
        b = prompt_embeds.shape[0]
        dev = prompt_embeds.device
        if self.graph and self._graphs:
            tag = self._weights_signature()
            if tag != self._weights_tag:                 # a parameter was updated / reloaded since the captures were made
                self._graphs.clear()
            self._weights_tag = tag
        elif self.graph:
            self._weights_tag = self._weights_signature()
        cfg = guidance_scale > 1.0 and negative_prompt_embeds is not None
        gdev = generator.device if generator is not None else "cpu"
        if latents is None:
            shape = (b, self.unet.config.in_channels, num_frames, height // 8, width // 8)
            latents = torch.randn(shape, generator=generator, device=gdev).to(dev)
        ehs = torch.cat([negative_prompt_embeds, prompt_embeds], 0) if cfg else prompt_embeds
        sch = self.scheduler
        timesteps = sch.set_timesteps(num_inference_steps)
        if init_weight > 0:                      # start from a partially noised input clip (inference.py:189-197)
            start = round(init_weight * len(timesteps))
            noise = torch.randn(latents.shape, generator=generator, device=gdev).to(dev)
            latents = sch.add_noise(latents, noise, timesteps[start:start + 1].to(dev))
            timesteps = timesteps[start:]
            sch._step_index = start
        nf = latents.shape[2]
        ws = min(nf, window_size) if window_size else nf
        if ws >= nf and not rotate:
            for t in timesteps:
                latents = sch.step(self._eps(latents, t, ehs, cfg, guidance_scale, dev), t, latents)
        else:
            latents = self._diffuse_windows(latents, timesteps, ehs, cfg, guidance_scale, dev, ws, rotate, generator)
        if decode:
            if self.vae is None:
                raise RuntimeError("t2v_amd: decode=True needs a VAE (TextToVideoSampler(unet, scheduler, vae))")
            from .models.vae import decode_latents
            return decode_latents(latents, self.vae, vae_batch_size)
        return latents

    def _diffuse_windows(self, latents, timesteps, ehs, cfg, guidance_scale, dev, ws, rotate, generator):
        """inference.py:199-262.  The solver's per-frame history (`model_outputs` there; here the previous data prediction
        `_prev_x0`) lives in `hist` and is sliced / rolled together with the latents; the scalar part of the solver state
        (step index, previous log-SNR) is common to all windows of a timestep."""
        sch = self.scheduler
        nf = latents.shape[2]
        hist = None                                  # previous data prediction of every frame (multistep solvers)
        shifts, total_shift = [], 0
        if rotate:
            pr = primes_up_to(ws)
            perm = torch.randperm(len(pr), generator=generator, device=generator.device if generator is not None else "cpu")
            shifts = [pr[int(i)] for i in perm]
        for i, t in enumerate(timesteps):
            if rotate:
                sh = shifts[i % len(shifts)]
                latents = torch.roll(latents, shifts=sh, dims=2)
                hist = None if hist is None else torch.roll(hist, shifts=sh, dims=2)
                total_shift += sh
            step0, lam0 = sch._step_index, getattr(sch, "_prev_lambda", None)
            new_lat = torch.empty_like(latents)
            new_hist = None
            for s in range(0, nf, ws):
                sch._step_index, sch._prev_lambda = step0, lam0
                sch._prev_x0 = None if hist is None else hist[:, :, s:s + ws]
                win = latents[:, :, s:s + ws]
                new_lat[:, :, s:s + ws] = sch.step(self._eps(win, t, ehs, cfg, guidance_scale, dev), t, win)
                px = getattr(sch, "_prev_x0", None)
                if px is not None:
                    if new_hist is None:
                        new_hist = torch.empty_like(latents)
                    new_hist[:, :, s:s + ws] = px
            latents, hist = new_lat, new_hist
        if rotate:
            latents = torch.roll(latents, shifts=-total_shift, dims=2)
        return latents
