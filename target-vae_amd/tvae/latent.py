"""Inference-time latent extraction (reference clustering_mnist.py:45-164, attention/attention branch :121-161):
encoder forward on the HIP kernels, then ONE epilogue kernel (argmax over (r,h,w), gather, softmax-expected
translation) instead of the reference's ~20 ATen launches and per-call host grid rebuild."""
from __future__ import annotations

import torch

from . import step
from ._lib import call


def get_latent(x, y, encoder_model, t_inf, r_inf, device, image_dim):
    """Reference signature clustering_mnist.py:45.  Returns (z_content (B, 2z) = [z_mu, z_std], theta_mu (B,1),
    dx (B,2)); z_std = exp(logstd) without the training-time epsilon (reference :137)."""
    step._check_branch(t_inf, r_inf)
    with torch.no_grad():
        y = y.to(device)
        enc = encoder_model
        B, R, Ho, zd = y.shape[0], enc.groupconv, enc.output_size(), enc.latent_dim
        heads = enc.encode_heads(y)
        tb = enc.head_tables(y.device, step.pixel_spacing(x.to(device)))
        zc = torch.empty(B, 2 * zd, dtype=torch.float32, device=y.device)
        th = torch.empty(B, 1, dtype=torch.float32, device=y.device)
        dx = torch.empty(B, 2, dtype=torch.float32, device=y.device)
        call('tvae_get_latent', heads, heads.shape[1], tb.p_r, tb.off, tb.grid, B, R, Ho * Ho, zd, tb.theta_off_scale,
             zc, th, dx)
    return zc, th, dx
