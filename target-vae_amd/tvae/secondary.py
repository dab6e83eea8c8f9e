"""Secondary inference branches of eval_minibatch (reference train_mnist.py:35-185): `--t-inf unimodal --r-inf
unimodal` (MLP encoder) and `--t-inf attention --r-inf unimodal` (translation attention; rotation pooled by `fc_r`
over the HIP GroupConv, or a plain convolution with `--groupconv 0`).

They are not in any BASELINE configuration.  Both encoders, the coordinate transform, the decoder
and the likelihood run on the HIP kernels (ops.MlpFn, ops.TransAttnEncoderFn: SURVEY 8f row 4); only the small
posterior tail between them (reductions over (B, Ho^2) tensors) is generic torch.  Noise can be injected for parity
tests.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops, tables

LOG_SQRT_2PI = math.log(math.sqrt(2 * math.pi))


def _rotate(x, dx, theta):
    """x' = (x - dx) R(theta) on the HIP coordinate kernel (train_mnist.py:65-76)."""
    return ops.CoordFn.apply(x, dx.contiguous(), theta.contiguous())


def unimodal_unimodal(x, y, generator_model, encoder_model, theta_prior, likelihood='bce', eps=None):
    """train_mnist.py:35-83.  The first latent is the rotation, the next two the translation (scaled by 0.1)."""
    b = y.shape[0]
    z_mu, z_logstd = encoder_model(y.reshape(b, -1))
    z_std = torch.exp(z_logstd)
    if eps is None:
        eps = torch.randn_like(z_mu)
    z = z_std * eps + z_mu
    theta = z[:, 0]
    sigma = float(theta_prior)
    kl = -z_logstd[:, 0] + math.log(sigma) + (z_std[:, 0] ** 2 + z_mu[:, 0] ** 2) / 2 / sigma ** 2 - 0.5
    rest_mu, rest_std, rest_logstd = z_mu[:, 1:], z_std[:, 1:], z_logstd[:, 1:]
    dx = z[:, 1:3] * 0.1
    content = z[:, 3:]
    xr = _rotate(x, dx, theta)
    kl = kl + (-rest_logstd + 0.5 * rest_std ** 2 + 0.5 * rest_mu ** 2 - 0.5).sum(1)
    kl_div = kl.mean()
    y_hat = generator_model(xr, content)
    lp = ops.LogLikFn.apply(y_hat, y, ops.LIK_KIND[likelihood])
    log_p = lp.mean()
    return log_p - kl_div, log_p, kl_div


def attention_unimodal(x, y, generator_model, encoder_model, theta_prior, spacing, likelihood='bce', noise=None):
    """train_mnist.py:86-185: attention over translations only; q(theta | t) Gaussian with N(0, theta_prior) prior."""
    b = y.shape[0]
    dev = y.device
    E = eps_z = eps_t = None
    if noise is not None:
        E, eps_z, eps_t = noise
    attn, a_s, theta_vals, z_vals = encoder_model(y, dev, E=E)
    Ho = attn.shape[3]
    a = a_s.reshape(b, -1, 1)
    z_vals = z_vals.reshape(b, z_vals.shape[1], -1)
    theta_vals = theta_vals.reshape(b, 2, -1)
    zd = z_vals.shape[1] // 2
    z_mu, z_std = z_vals[:, :zd], torch.exp(z_vals[:, zd:]) + 1e-6
    if eps_z is None:
        eps_z = torch.randn(b, zd, device=dev)
        eps_t = torch.randn(b, device=dev)
    z = (torch.bmm(z_std, a) * eps_z.view(b, zd, 1) + torch.bmm(z_mu, a)).squeeze(2)
    G64 = torch.from_numpy(tables.translation_grid(Ho, spacing)).to(dev)             # float64 like the reference
    G = G64.float()
    dx = torch.bmm(G.t().expand(b, 2, -1), a).squeeze(2)
    th_mu, th_std = theta_vals[:, 0:1], torch.exp(theta_vals[:, 1:2]) + 1e-6
    theta = (torch.bmm(th_std, a) * eps_t.view(b, 1, 1) + torch.bmm(th_mu, a)).reshape(b)
    xr = _rotate(x, dx, theta)
    q_t = F.log_softmax(attn.reshape(b, -1), dim=1)
    dead = torch.exp(q_t) == 0                                                        # NaN guards, :153-162
    zm = torch.where(dead.unsqueeze(1), torch.zeros_like(z_mu), z_mu)
    zs = torch.where(dead.unsqueeze(1), torch.ones_like(z_std), z_std)
    tm = torch.where(dead, torch.zeros_like(q_t), th_mu.squeeze(1))
    ts = torch.where(dead, torch.ones_like(q_t), th_std.squeeze(1))
    sd = float(np.float32(0.1))
    p_t = (-(G64 ** 2) / (2 * sd * sd) - math.log(sd) - LOG_SQRT_2PI).sum(1)
    p_t = F.log_softmax(p_t, dim=0).unsqueeze(0)
    eq = torch.exp(q_t)
    val1 = (eq * (q_t - p_t)).sum(1)
    kl_z = (0.5 * (zs ** 2 + zm ** 2 - 1.0 - torch.log(zs ** 2))).sum(1)
    sp = float(np.float32(theta_prior))
    kl_t = 0.5 * ((ts / sp) ** 2 + (tm / sp) ** 2 - 1.0 - torch.log((ts / sp) ** 2))
    val2 = (eq * (kl_t + kl_z)).sum(1)
    kl_div = (val1 + val2).mean()
    y_hat = generator_model(xr, z)
    lp = ops.LogLikFn.apply(y_hat, y, ops.LIK_KIND[likelihood])
    log_p = lp.mean()
    return log_p - kl_div, log_p, kl_div
