"""DDPM forward process used by the train step (`noise_scheduler.add_noise` train.py:760, `get_velocity` :797).
ModelScope scheduler config: scaled_linear 0.00085 -> 0.012, 1000 steps, epsilon prediction.
Elementwise math on the (B,4,F,h,w) latent — a few KB — stays in torch."""
import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="epsilon"):
        if beta_schedule != "scaled_linear":
            raise ValueError("only scaled_linear is used by the reference checkpoints")
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.config = type("cfg", (), dict(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type))()
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self._acp_cache = {}

    def _coeffs(self, like, timesteps):
        key = (like.device, like.dtype)
        acp = self._acp_cache.get(key)
        if acp is None:     # cached so that a HIP-graph capture never sees a host->device copy
            acp = self._acp_cache[key] = self.alphas_cumprod.to(device=like.device, dtype=like.dtype)
        a = acp[timesteps] ** 0.5
        s = (1 - acp[timesteps]) ** 0.5
        while a.dim() < like.dim():
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a, s

    def add_noise(self, original_samples, noise, timesteps):
        a, s = self._coeffs(original_samples, timesteps)
        return a * original_samples + s * noise

    def get_velocity(self, sample, noise, timesteps):
        a, s = self._coeffs(sample, timesteps)
        return a * noise - s * sample


def enforce_zero_terminal_snr(betas):
    """train.py:360-389 ("Common Diffusion Noise Schedules and Sample Steps are Flawed"): shift/scale sqrt(alpha_bar) so the
    last timestep has zero SNR.  As in the reference (`train.py:689-690`) the result is only assigned to `.betas`; the
    cached `alphas_cumprod` used by `add_noise` is NOT recomputed there, so `rescale_schedule` does not change training
    noise levels — `DDPMScheduler.rescale_betas()` reproduces that behaviour."""
    abar_sqrt = (1 - betas).cumprod(0).sqrt()
    a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
    abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
    abar = abar_sqrt ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas


def _rescale_betas(self):
    self.betas = enforce_zero_terminal_snr(self.betas)


DDPMScheduler.rescale_betas = _rescale_betas
