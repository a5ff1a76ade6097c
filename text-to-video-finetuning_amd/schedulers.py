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


# --------------------------------------------------------------------------- sampling schedulers
class _SamplerBase(DDPMScheduler):
    """Shared timestep grid of the sampling schedulers: `num_inference_steps` points of [0, T-1], descending
    ("linspace" spacing).  alpha_t = sqrt(abar_t), sigma_t = sqrt(1 - abar_t); the step past the last point is abar = 1."""

    def set_timesteps(self, num_inference_steps, device=None):
        ts = torch.linspace(0, self.num_train_timesteps - 1, num_inference_steps + 1).round().long().flip(0)[:-1]
        self.timesteps = ts.to(device) if device is not None else ts
        self.num_inference_steps = num_inference_steps
        self._step_index = 0
        self._prev_x0 = None
        self._prev_lambda = None
        return self.timesteps

    def _abar(self, t):
        return self.alphas_cumprod[t].double() if t >= 0 else torch.tensor(1.0, dtype=torch.float64)

    def _x0(self, model_output, sample, t):
        a, s = self._abar(t).sqrt(), (1 - self._abar(t)).sqrt()
        if self.prediction_type == "epsilon":
            return (sample - s.to(sample.dtype) * model_output) / a.to(sample.dtype)
        if self.prediction_type == "v_prediction":
            return a.to(sample.dtype) * sample - s.to(sample.dtype) * model_output
        raise ValueError(f"Unknown prediction type {self.prediction_type}")

    def _next_t(self):
        i = self._step_index
        return int(self.timesteps[i + 1]) if i + 1 < len(self.timesteps) else -1


class DDIMScheduler(_SamplerBase):
    """Deterministic DDIM (eta = 0): x_prev = alpha_prev * x0 + sigma_prev * eps, with (x0, eps) recovered from the model
    output at t.  Used by `inference.py` style sampling; exact along the true trajectory for an exact model.
    `timestep_spacing="leading"`, `steps_offset=1`, `set_alpha_to_one=False` reproduce the grid of the ModelScope
    `scheduler_config.json` (t_i = i * (T // n) + offset, previous step t - T // n, abar past the end = abar_0); the default
    is the "linspace" grid shared with the DPM solver."""

    def __init__(self, *args, timestep_spacing="linspace", steps_offset=0, set_alpha_to_one=True, **kw):
        super().__init__(*args, **kw)
        self.timestep_spacing, self.steps_offset, self.set_alpha_to_one = timestep_spacing, steps_offset, set_alpha_to_one

    def set_timesteps(self, num_inference_steps, device=None):
        if self.timestep_spacing != "leading":
            return super().set_timesteps(num_inference_steps, device)
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (torch.arange(num_inference_steps) * ratio).flip(0).long() + self.steps_offset
        self.timesteps = ts.to(device) if device is not None else ts
        self.num_inference_steps, self._ratio = num_inference_steps, ratio
        self._step_index, self._prev_x0, self._prev_lambda = 0, None, None
        return self.timesteps

    def _abar(self, t):
        if t >= 0:
            return self.alphas_cumprod[t].double()
        return torch.tensor(1.0, dtype=torch.float64) if self.set_alpha_to_one else self.alphas_cumprod[0].double()

    def _next_t(self):
        if self.timestep_spacing == "leading":
            return int(self.timesteps[self._step_index]) - self._ratio
        return super()._next_t()

    def step(self, model_output, timestep, sample):
        t, tp = int(timestep), self._next_t()
        x0 = self._x0(model_output, sample, t)
        a_t, s_t = self._abar(t).sqrt(), (1 - self._abar(t)).sqrt()
        eps = (sample - a_t.to(sample.dtype) * x0) / s_t.to(sample.dtype)
        a_p, s_p = self._abar(tp).sqrt(), (1 - self._abar(tp)).sqrt()
        self._step_index += 1
        return a_p.to(sample.dtype) * x0 + s_p.to(sample.dtype) * eps


class DPMSolverMultistepScheduler(_SamplerBase):
    """DPM-Solver++(2M), data-prediction form, midpoint variant, first-order warm-up; the final step (sigma = 0) lands on the
    data prediction — the sampler the reference's validation uses (`DPMSolverMultistepScheduler.from_config`, train.py:923-926;
    algorithm of Lu et al. 2022, eq. for the multistep second-order update):
        lambda_t = log(alpha_t / sigma_t), h = lambda_t - lambda_s
        first  : x_t = (sigma_t/sigma_s) x_s - alpha_t (e^{-h} - 1) D_s
        second : D1 = (D_s - D_s') / r0, r0 = (lambda_s - lambda_s') / h ;  x_t = first - 0.5 alpha_t (e^{-h} - 1) D1"""

    def __init__(self, *args, solver_order=2, lower_order_final=True, **kw):
        super().__init__(*args, **kw)
        self.solver_order, self.lower_order_final = solver_order, lower_order_final

    def step(self, model_output, timestep, sample):
        t, tp = int(timestep), self._next_t()
        x0 = self._x0(model_output, sample, t)
        a_s, s_s = self._abar(t).sqrt(), (1 - self._abar(t)).sqrt()
        a_t, s_t = self._abar(tp).sqrt(), (1 - self._abar(tp)).sqrt()
        lam_s = torch.log(a_s / s_s)
        last = tp < 0
        if last:                                         # sigma_t = 0: the update collapses onto the data prediction
            out = x0
        else:
            lam_t = torch.log(a_t / s_t)
            h = lam_t - lam_s
            em1 = torch.expm1(-h)
            out = (s_t / s_s).to(sample.dtype) * sample - (a_t * em1).to(sample.dtype) * x0
            # diffusers' `lower_order_final` (schedules < 15 steps) makes only the LAST step first order; with a final
            # sigma of 0 that step already collapses onto the data prediction above, so every earlier step stays second order
            second = self.solver_order >= 2 and self._prev_x0 is not None
            if second:
                r0 = (lam_s - self._prev_lambda) / h
                d1 = (x0 - self._prev_x0) / r0.to(sample.dtype)
                out = out - (0.5 * a_t * em1).to(sample.dtype) * d1
        self._prev_x0, self._prev_lambda = x0, lam_s
        self._step_index += 1
        return out
