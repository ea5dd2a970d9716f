"""Noise schedules and the per-timestep coefficient buffers of the reverse process.

Host-side, runs once at construction.  Arithmetic is done in float64 with torch CPU ops and
cast to float32 at the end, exactly like the reference (/root/reference/ddpm.py:460-494 for the
three beta schedules, :547-593 for the derived buffers, :567 for the fp64->fp32 cast), so the
buffers are bit-identical to a reference ``GaussianDiffusion``'s registered buffers.
"""
import math
from collections import OrderedDict

import torch

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev",
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
    "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2", "loss_weight",
)


def _grid(timesteps):
    return torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps


def betas_linear(timesteps):
    k = 1000.0 / timesteps                                   # ddpm.py:464-467
    return torch.linspace(k * 1e-4, k * 2e-2, timesteps, dtype=torch.float64)


def _betas_from_abar(abar):
    abar = abar / abar[0]
    return torch.clip(1.0 - abar[1:] / abar[:-1], 0, 0.999)


def betas_cosine(timesteps, s=0.008):
    t = _grid(timesteps)                                     # ddpm.py:474-479
    return _betas_from_abar(torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2)


def betas_sigmoid(timesteps, start=-3, end=3, tau=1, clamp_min=1e-5):
    t = _grid(timesteps)                                     # ddpm.py:487-494
    lo = torch.tensor(start / tau).sigmoid()
    hi = torch.tensor(end / tau).sigmoid()
    abar = (hi - ((t * (end - start) + start) / tau).sigmoid()) / (hi - lo)
    return _betas_from_abar(abar)


_SCHEDULES = {"linear": betas_linear, "cosine": betas_cosine, "sigmoid": betas_sigmoid}


def make_buffers(timesteps, beta_schedule="sigmoid", objective="pred_x0",
                 min_snr_loss_weight=False, min_snr_gamma=5, **schedule_kwargs):
    """-> OrderedDict name -> float32 [T] tensor (the reference's 13 registered buffers)."""
    if beta_schedule not in _SCHEDULES:
        raise ValueError(f"unknown beta schedule {beta_schedule}")
    beta = _SCHEDULES[beta_schedule](timesteps, **schedule_kwargs)
    alpha = 1.0 - beta
    abar = torch.cumprod(alpha, dim=0)
    abar_prev = torch.cat([torch.ones(1, dtype=torch.float64), abar[:-1]])
    post_var = beta * (1.0 - abar_prev) / (1.0 - abar)
    snr = abar / (1 - abar)
    clipped = snr.clone()
    if min_snr_loss_weight:
        clipped.clamp_(max=min_snr_gamma)
    if objective == "pred_noise":
        lw = clipped / snr
    elif objective == "pred_x0":
        lw = clipped
    elif objective == "pred_v":
        lw = clipped / (snr + 1)
    else:
        raise ValueError(f"unknown objective {objective}")
    vals = OrderedDict(
        betas=beta, alphas_cumprod=abar, alphas_cumprod_prev=abar_prev,
        sqrt_alphas_cumprod=abar.sqrt(),
        sqrt_one_minus_alphas_cumprod=(1.0 - abar).sqrt(),
        log_one_minus_alphas_cumprod=(1.0 - abar).log(),
        sqrt_recip_alphas_cumprod=(1.0 / abar).sqrt(),
        sqrt_recipm1_alphas_cumprod=(1.0 / abar - 1).sqrt(),
        posterior_variance=post_var,
        posterior_log_variance_clipped=post_var.clamp(min=1e-20).log(),
        posterior_mean_coef1=beta * abar_prev.sqrt() / (1.0 - abar),
        posterior_mean_coef2=(1.0 - abar_prev) * alpha.sqrt() / (1.0 - abar),
        loss_weight=lw,
    )
    return OrderedDict((k, v.to(torch.float32)) for k, v in vals.items())


def ddim_time_pairs(total_timesteps, sampling_timesteps):
    """[(t, t_next), ...] as the reference builds them (/root/reference/ddpm.py:984-986)."""
    times = torch.linspace(-1, total_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return times, list(zip(times[:-1], times[1:]))
