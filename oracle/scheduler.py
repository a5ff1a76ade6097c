"""DDPM forward-process pieces used by the train step (ORACLE — test infrastructure).

`noise_scheduler.add_noise` at `train.py:760` and `get_velocity` at `train.py:797` are
un-vendored diffusers `DDPMScheduler` methods; the ModelScope
scheduler config is scaled_linear 0.00085 -> 0.012, 1000 steps, epsilon prediction.
`enforce_zero_terminal_snr` restates `train.py:360-389`.
"""
import torch


def scaled_linear_betas(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2


def alphas_cumprod(betas=None):
    betas = scaled_linear_betas() if betas is None else betas
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(original, noise, timesteps, acp=None):
    acp = alphas_cumprod() if acp is None else acp
    acp = acp.to(original.device, original.dtype)
    a = acp[timesteps] ** 0.5
    s = (1 - acp[timesteps]) ** 0.5
    while a.dim() < original.dim():
        a, s = a.unsqueeze(-1), s.unsqueeze(-1)
    return a * original + s * noise


def get_velocity(sample, noise, timesteps, acp=None):
    acp = alphas_cumprod() if acp is None else acp
    acp = acp.to(sample.device, sample.dtype)
    a = acp[timesteps] ** 0.5
    s = (1 - acp[timesteps]) ** 0.5
    while a.dim() < sample.dim():
        a, s = a.unsqueeze(-1), s.unsqueeze(-1)
    return a * noise - s * sample


def enforce_zero_terminal_snr(betas):
    """train.py:360-389: shift/scale sqrt(alpha_bar) so the terminal SNR is zero."""
    abar_sqrt = (1 - betas).cumprod(0).sqrt()
    a0, aT = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
    abar_sqrt = (abar_sqrt - aT) * (a0 / (a0 - aT))
    abar = abar_sqrt ** 2
    alphas = torch.cat([abar[0:1], abar[1:] / abar[:-1]])
    return 1 - alphas
