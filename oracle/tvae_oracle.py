"""CPU oracle for the TARGET-VAE training hot path.

TEST INFRASTRUCTURE ONLY.  This module is a plain PyTorch-CPU fp32 restatement of the
reference algorithm (SMLC-NYSBC/TARGET-VAE, mounted read-only at /root/reference in the
build container).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / reported baseline.  Nothing under
`target-vae_amd/` (the product) imports or executes anything in `oracle/`.

Parity pin: every function here is checked against outputs of the real reference, imported
in the build container by `tests/golden/make_goldens.py`, and against the committed
fixtures `tests/golden/*.npz` (see tests/test_oracle_golden.py).  Parity is PINNED by
those fixtures (the reference itself ships no tests or golden vectors).

Each function cites the reference file:line it restates.  All randomness is injected
explicitly (E ~ Exp(1) for the Gumbel-softmax, eps_z, eps_theta ~ N(0,1)); the reference
draws them in that order per step (models.py:387, train_mnist.py:206, :230).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LRELU_SLOPE = 0.01  # nn.LeakyReLU default, models.py:327 (activation=nn.LeakyReLU)
EPS_STD = 1e-6      # train_mnist.py:197
DX_PRIOR_STD = 0.1  # train_mnist.py:258


# --------------------------------------------------------------------------------------
# constants that depend only on shapes
# --------------------------------------------------------------------------------------
def rotation_offsets(R: int) -> np.ndarray:
    """Per-rotation angle offsets, models.py:361-366 (tables for R in {4,8,16}).

    o_r = r*pi/(R/2) for r <= R/2, (r-R)*pi/(R/2) otherwise; float32 like `.type(torch.float)`.
    """
    half = R // 2
    vals = [(r * np.pi / half) if r <= half else ((r - R) * np.pi / half) for r in range(R)]
    return np.asarray(vals, dtype=np.float64).astype(np.float32)


def rotation_log_prior(R: int, rot_refinement: bool, theta_prior: float,
                       normal_prior_over_r: bool) -> np.ndarray:
    """log p(r), models.py:368-379.  Returns float32 (R,)."""
    if rot_refinement:
        off = torch.from_numpy(rotation_offsets(R))
        if normal_prior_over_r:
            d = torch.distributions.Normal(torch.tensor([0.0]), torch.tensor([float(theta_prior)]))
        else:
            d = torch.distributions.Uniform(torch.tensor([-2 * np.pi]), torch.tensor([2 * np.pi]))
        return d.log_prob(off).numpy().astype(np.float32)
    return (torch.zeros(R) - np.log(R)).numpy().astype(np.float32)


def translation_grid(Ho: int, spacing: float) -> np.ndarray:
    """float64 (Ho*Ho, 2) grid of candidate translations, train_mnist.py:209-218.

    `spacing` is the float32 pixel spacing x_coord[1,0]-x_coord[0,0] (train_mnist.py:30).
    Built from integer index * spacing (equals np.arange for the Ho used here).
    """
    s = np.float64(np.float32(spacing))
    half = Ho // 2
    if Ho % 2:
        g = (np.arange(Ho, dtype=np.float64) - half) * s
    else:
        g = (np.arange(Ho, dtype=np.float64) - half) * s
    x0, x1 = np.meshgrid(g, g[::-1])
    return np.stack([x0.ravel(), x1.ravel()], 1)


def image_coords(n: int) -> Tensor:
    """x_coord (n*n, 2) float32, train_mnist.py:475-479."""
    xg = np.linspace(-1, 1, n)
    yg = np.linspace(1, -1, n)
    x0, x1 = np.meshgrid(xg, yg)
    return torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1)).float()


# --------------------------------------------------------------------------------------
# GroupConv  (models.py:132-225)
# --------------------------------------------------------------------------------------
def rotation_taps(k: int, R: int) -> Tuple[np.ndarray, np.ndarray]:
    """Bilinear taps of GroupConv.trans_filter (models.py:174-197) in its 2-D form.

    For rotation r and output pixel (u,v) returns 4 source indices into the flattened
    k*k filter (or -1 if out of range) and 4 float32 weights, following
    F.affine_grid/F.grid_sample with align_corners=False and zeros padding, evaluated in
    float32 like ATen does.  theta_r is accumulated in float64 (models.py:195).
    """
    idx = np.full((R, k * k, 4), -1, dtype=np.int64)
    wgt = np.zeros((R, k * k, 4), dtype=np.float32)
    lin = (np.linspace(-1, 1, k, dtype=np.float32) * np.float32((k - 1) / k)).astype(np.float32)
    xt, yt = np.meshgrid(lin, lin)  # xt varies along last dim (v), yt along u
    d_theta = 2 * np.pi / R
    theta = 0.0
    for r in range(R):
        c = np.float32(np.cos(theta))
        s = np.float32(np.sin(theta))
        ms = np.float32(-np.sin(theta))
        gx = (c * xt + s * yt).astype(np.float32)
        gy = (ms * xt + c * yt).astype(np.float32)
        ix = ((gx + np.float32(1)) * np.float32(k) - np.float32(1)) / np.float32(2)
        iy = ((gy + np.float32(1)) * np.float32(k) - np.float32(1)) / np.float32(2)
        ix0 = np.floor(ix)
        iy0 = np.floor(iy)
        wx1 = (ix - ix0).astype(np.float32)
        wy1 = (iy - iy0).astype(np.float32)
        wx0 = (np.float32(1) - wx1).astype(np.float32)
        wy0 = (np.float32(1) - wy1).astype(np.float32)
        ix0 = ix0.astype(np.int64)
        iy0 = iy0.astype(np.int64)
        t = 0
        for dy, wy in ((0, wy0), (1, wy1)):
            for dx, wx in ((0, wx0), (1, wx1)):
                yy = iy0 + dy
                xx = ix0 + dx
                ok = (yy >= 0) & (yy < k) & (xx >= 0) & (xx < k)
                flat = np.where(ok, yy * k + xx, -1)
                idx[r, :, t] = flat.ravel()
                wgt[r, :, t] = np.where(ok, wy * wx, np.float32(0)).astype(np.float32).ravel()
                t += 1
        theta += d_theta
    return idx, wgt


def rotated_bank(weight: Tensor, R: int) -> Tensor:
    """GroupConv.trans_filter (models.py:174-197): (C,Cin,1,k,k) -> (C,R,Cin,1,k,k)."""
    C, Cin, D, k, _ = weight.shape
    assert D == 1, "input_rot_dim is always 1 in the reference (models.py:290,346)"
    idx, wgt = rotation_taps(k, R)
    flat = weight.reshape(C, Cin, k * k)
    zero = torch.zeros(C, Cin, 1, dtype=weight.dtype)
    flat0 = torch.cat([flat, zero], dim=2)  # index k*k -> 0 for out-of-range taps
    idx_t = torch.from_numpy(np.where(idx < 0, k * k, idx))
    wgt_t = torch.from_numpy(wgt)
    out = []
    for r in range(R):
        acc = 0
        for t in range(4):
            acc = acc + flat0[:, :, idx_t[r, :, t]] * wgt_t[r, :, t]
        out.append(acc.view(C, Cin, 1, k, k))
    return torch.stack(out, dim=1)


def groupconv_forward(x: Tensor, weight: Tensor, bias: Optional[Tensor], R: int, padding: int,
                      stride: int = 1) -> Tensor:
    """GroupConv.forward (models.py:202-225): (B,Cin,n,n) -> (B,C,R,Ho,Ho)."""
    C, Cin, _, k, _ = weight.shape
    tw = rotated_bank(weight, R).view(C * R, Cin, k, k)
    y = F.conv2d(x.view(x.shape[0], Cin, x.shape[-2], x.shape[-1]), tw, None, stride, padding)
    B, _, ho, wo = y.shape
    y = y.view(B, C, R, ho, wo)
    if bias is not None:
        y = y + bias.view(1, C, 1, 1, 1)
    return y


# --------------------------------------------------------------------------------------
# encoder  (models.py:326-403)
# --------------------------------------------------------------------------------------
def lrelu(x: Tensor) -> Tensor:
    return F.leaky_relu(x, LRELU_SLOPE)


def encoder_forward(p: Dict[str, Tensor], y: Tensor, E: Tensor, R: int, padding: int,
                    rot_refinement: bool, theta_prior: float = np.pi,
                    normal_prior_over_r: bool = True):
    """InferenceNetwork_AttentionTranslation_AttentionRotation.forward (models.py:354-403).

    `p` is the module's state_dict (conv1.weight, conv1.bias, conv2.*, conv_a.*, conv_r.*,
    conv_z.*).  `E` (B, R*Ho*Ho) are the Exp(1) draws of F.gumbel_softmax (models.py:387:
    gumbels = -empty_like(logits).exponential_().log()).
    Returns the reference 7-tuple (attn, q_t_r, p_r, a_sampled, offsets, theta, z).
    """
    C = p['conv1.weight'].shape[0]
    x = lrelu(groupconv_forward(y, p['conv1.weight'], p['conv1.bias'], R, padding))   # :355
    B, _, _, Ho, Wo = x.shape

    def pw(t, w, b):  # Conv3d(.,.,1) == per-position channel GEMM
        return torch.einsum('oc,bcrhw->borhw', w.view(w.shape[0], -1), t) + b.view(1, -1, 1, 1, 1)

    h = lrelu(pw(x, p['conv2.weight'], p['conv2.bias']))                                # :356
    attn = pw(h, p['conv_a.weight'], p['conv_a.bias']).squeeze(1)                       # :358
    p_r = torch.from_numpy(rotation_log_prior(R, rot_refinement, theta_prior,
                                              normal_prior_over_r)).view(R, 1, 1)       # :360-379
    attn = attn + p_r                                                                   # :382
    q_t_r = F.log_softmax(attn.reshape(B, -1), dim=1).view(B, R, Ho, Wo)                # :383
    a = attn.reshape(B, -1)
    g = a - torch.log(E.view(B, -1))                                                    # :387 (tau=1)
    a_sampled = F.softmax(g, dim=-1).view(B, R, Ho, Wo)                                 # :387-388
    z = pw(h, p['conv_z.weight'], p['conv_z.bias'])                                     # :390
    theta = pw(h, p['conv_r.weight'], p['conv_r.bias'])                                 # :392
    if rot_refinement:
        offsets = torch.from_numpy(rotation_offsets(R))
        theta_mu = theta[:, 0] + offsets.view(1, R, 1, 1)                               # :394-397
        theta = torch.stack((theta_mu, theta[:, 1]), dim=1)                             # :399
    else:
        offsets = torch.zeros(R)                                                        # :401
    return attn, q_t_r, p_r, a_sampled, offsets, theta, z


def encoder_heads(p: Dict[str, Tensor], y: Tensor, R: int, padding: int):
    """The deterministic part of the encoder forward up to the three 1x1x1 heads (models.py:355-358,390,392), for
    tests that need the pre-activations: returns (heads, pre1, pre2) with heads (B, 3+2z, R, Ho, Ho) stacked as
    (conv_a | conv_r (2) | conv_z (2z)) BEFORE the prior / offsets are added, pre1 = conv1 output + bias (input of the
    first LeakyReLU, :355) and pre2 = conv2 output (input of the second, :356)."""
    def pw(t, w, b):
        return torch.einsum('oc,bcrhw->borhw', w.view(w.shape[0], -1), t) + b.view(1, -1, 1, 1, 1)
    pre1 = groupconv_forward(y, p['conv1.weight'], p['conv1.bias'], R, padding)
    pre2 = pw(lrelu(pre1), p['conv2.weight'], p['conv2.bias'])
    h = lrelu(pre2)
    heads = torch.cat([pw(h, p['conv_a.weight'], p['conv_a.bias']), pw(h, p['conv_r.weight'], p['conv_r.bias']),
                       pw(h, p['conv_z.weight'], p['conv_z.bias'])], dim=1)
    return heads, pre1, pre2


# --------------------------------------------------------------------------------------
# SpatialGenerator  (models.py:65-123)
# --------------------------------------------------------------------------------------
def generator_forward(g: Dict[str, Tensor], x: Tensor, z: Optional[Tensor], num_layers: int,
                      resid: bool = False, fourier_sigma: Optional[float] = None) -> Tensor:
    """SpatialGenerator.forward (models.py:95-123) for LeakyReLU activations.

    `g` is the module's state_dict; Fourier buffers embed_latent.{weight,bias} are used when
    `fourier_sigma` is given (sigma is a plain attribute, models.py:40).  x (B,N,2), z (B,zdim).
    """
    if x.dim() < 3:
        x = x.unsqueeze(0)
    b, n = x.shape[0], x.shape[1]
    h = x.reshape(b * n, -1)
    if fourier_sigma is not None:
        sig = torch.tensor(fourier_sigma, dtype=torch.float32)
        h = torch.cos(F.linear(h, g['embed_latent.weight'] / sig, g['embed_latent.bias']))  # :53-58
    h = F.linear(h, g['coord_linear.weight'], g['coord_linear.bias']).view(b, n, -1)        # :107
    if 'latent_linear.weight' in g:
        zz = z if z.dim() >= 2 else z.unsqueeze(0)
        h = h + F.linear(zz, g['latent_linear.weight']).unsqueeze(1)                         # :111-116
    h = h.view(b * n, -1)
    h = lrelu(h)                                                                             # layers[0]
    li = 1
    for _ in range(1, num_layers):
        if resid:
            h = lrelu(F.linear(h, g[f'layers.{li}.linear.weight'], g[f'layers.{li}.linear.bias']) + h)
            li += 1
        else:
            h = lrelu(F.linear(h, g[f'layers.{li}.weight'], g[f'layers.{li}.bias']))
            li += 2
    yv = F.linear(h, g[f'layers.{li}.weight'], g[f'layers.{li}.bias'])
    return yv.view(b, n, -1)


# --------------------------------------------------------------------------------------
# ELBO step  (train_mnist.py:26-294, attention / attention(+offsets) branch :187-282)
# --------------------------------------------------------------------------------------
def elbo_step(x_coord: Tensor, y: Tensor, enc: Dict[str, Tensor], gen: Dict[str, Tensor], *,
              R: int, padding: int, rot_refinement: bool, theta_prior: float,
              normal_prior_over_r: bool, num_layers: int, resid: bool = False,
              fourier_sigma: Optional[float] = None, likelihood: str = 'bce',
              E: Tensor, eps_z: Tensor, eps_theta: Tensor, return_aux: bool = False,
              ctf: Optional[Tensor] = None, mask_radius: int = 0):
    """eval_minibatch, attention-translation / attention-rotation branch.

    likelihood: 'bce' (train_mnist.py:286-291), 'bce3' (train_galaxy.py:287-292),
    'gauss' (train_particles.py:338), 'gauss_var' (train_particles.py:293-296,336).
    Returns (elbo f64, log_p f32, kl f64) like the reference [+ aux dict].
    """
    b = y.shape[0]
    spacing = (x_coord[1, 0] - x_coord[0, 0]).numpy()                                   # :30
    x = x_coord.expand(b, x_coord.shape[0], x_coord.shape[1])                           # :31
    attn, q_t_r, p_r, a_s, offsets, theta_vals, z_vals = encoder_forward(
        enc, y, E, R, padding, rot_refinement, theta_prior, normal_prior_over_r)        # :190
    z, theta, dx, x, kl_per_image = posterior_pool_kl(x, attn, q_t_r, p_r, a_s, offsets, theta_vals, z_vals,
                                                      float(spacing), eps_z, eps_theta, R, theta_prior)
    kl_div = kl_per_image.mean()                                                        # :281-282

    y_hat = generator_forward(gen, x.contiguous(), z, num_layers, resid, fourier_sigma)  # :287
    if ctf is not None or mask_radius > 0:
        log_p = particles_logp(y_hat, y, ctf, mask_radius, dx, float(spacing))
    else:
        log_p = likelihood_logp(y_hat, y, likelihood)
    elbo = log_p - kl_div
    if return_aux:
        aux = dict(attn=attn, q_t_r=q_t_r, a_sampled=a_s, theta_vals=theta_vals, z_vals=z_vals,
                   z=z, theta=theta, dx=dx.view(b, 2), x_rot=x, y_hat=y_hat,
                   kl_per_image=kl_per_image)
        return elbo, log_p, kl_div, aux
    return elbo, log_p, kl_div


def posterior_pool_kl(x, attn, q_t_r, p_r, a_s, offsets, theta_vals, z_vals, spacing, eps_z, eps_theta, R,
                      theta_prior=np.pi):
    """Pooling, sampling, coordinate transform and KL of eval_minibatch (train_mnist.py:192-282), given the
    encoder 7-tuple.  x (b,N,2) expanded coordinates.  Returns z (b,zd), theta (b,), dx (b,1,2), rotated
    coordinates (b,N,2) and the per-image KL (float64, b)."""
    b = attn.shape[0]
    Ho = attn.shape[3]
    a_over_locs = a_s.sum(dim=1).view(b, -1, 1)                                         # :192
    a_flat = a_s.reshape(b, -1).unsqueeze(2)                                            # :193
    z_vals = z_vals.reshape(b, z_vals.shape[1], -1)
    theta_vals = theta_vals.reshape(b, theta_vals.shape[1], -1)
    zd = z_vals.shape[1] // 2
    z_mu = z_vals[:, :zd]
    z_std = torch.exp(z_vals[:, zd:]) + EPS_STD                                         # :199-202
    z_mu_e = torch.bmm(z_mu, a_flat)
    z_std_e = torch.bmm(z_std, a_flat)                                                  # :203-204
    z = (z_std_e * eps_z.view(b, zd, 1) + z_mu_e).squeeze(2)                            # :206-207

    G = torch.from_numpy(translation_grid(Ho, float(spacing)))                          # f64, :209-218
    Gb = G.expand(b, G.shape[0], 2).transpose(1, 2)
    dx = torch.bmm(Gb.type(torch.float), a_over_locs).squeeze(2).unsqueeze(1)           # :221
    x = x - dx                                                                          # :222

    theta_mu = theta_vals[:, 0:1]
    theta_std = torch.exp(theta_vals[:, 1:2]) + EPS_STD                                 # :225-227
    th_mu_e = torch.bmm(theta_mu, a_flat)
    th_std_e = torch.bmm(theta_std, a_flat)
    theta = (th_std_e * eps_theta.view(b, 1, 1) + th_mu_e).squeeze(2).squeeze(1)        # :230-231
    rot = torch.stack([torch.stack([torch.cos(theta), torch.sin(theta)], 1),
                       torch.stack([-torch.sin(theta), torch.cos(theta)], 1)], 1)       # :234-238
    x = torch.bmm(x, rot)                                                               # :239

    sh = (b, zd, R, Ho, Ho)
    q_tmp = q_t_r.unsqueeze(1).expand(*sh)
    z_mu5 = torch.where(torch.exp(q_tmp) == 0, torch.zeros_like(q_tmp), z_mu.reshape(*sh))     # :246
    z_std5 = torch.where(torch.exp(q_tmp) == 0, torch.ones_like(q_tmp), z_std.reshape(*sh))    # :247
    th_mu4 = torch.where(torch.exp(q_t_r) == 0, torch.zeros_like(q_t_r), theta_mu.reshape(b, R, Ho, Ho))
    th_std4 = torch.where(torch.exp(q_t_r) == 0, torch.ones_like(q_t_r), theta_std.reshape(b, R, Ho, Ho))

    p_t = torch.distributions.Normal(torch.tensor([0.0]), torch.tensor([DX_PRIOR_STD])) \
        .log_prob(G).sum(1).view(Ho, Ho).unsqueeze(0).unsqueeze(1)                      # f64, :258-259
    p_t_r = F.log_softmax((p_t + p_r.unsqueeze(0)).view(-1), dim=0).view(1, R, Ho, Ho)  # :261-262
    val1 = (torch.exp(q_t_r) * (q_t_r - p_t_r)).view(b, -1).sum(1)                      # :264 (f64)
    kl_z = (0.5 * (z_std5 ** 2 + z_mu5 ** 2 - 1.0 - torch.log(z_std5 ** 2))).sum(1)     # :266-267
    sp = np.pi / R if R >= 1 else theta_prior                                           # :269-272
    sp_t = torch.tensor([sp] * R).view(R, 1, 1)          # float32 like the reference tensor
    off = offsets.view(R, 1, 1)
    var_ratio = (th_std4 / sp_t) ** 2
    t1 = ((th_mu4 - off) / sp_t) ** 2
    kl_theta = 0.5 * (var_ratio + t1 - 1.0 - torch.log(var_ratio))                      # :274-276
    val2 = (torch.exp(q_t_r) * (kl_theta + kl_z)).view(b, -1).sum(1)                    # :278-279

    return z, theta, dx, x, val1 + val2


def likelihood_logp(y_hat: Tensor, y: Tensor, kind: str) -> Tensor:
    b = y.shape[0]
    if kind == 'bce':                                           # train_mnist.py:288-291
        yh = y_hat.reshape(b, -1)
        yy = y.reshape(b, -1)
        return -F.binary_cross_entropy_with_logits(yh, yy) * yy.shape[1]
    if kind == 'bce3':                                          # train_galaxy.py:288-292
        yh = y_hat.reshape(b, -1, 3)
        yy = y.reshape(b, -1, 3)
        return -F.binary_cross_entropy_with_logits(yh, yy) * (yy.shape[1] * 3)
    if kind == 'gauss':                                         # train_particles.py:284,338
        yh = y_hat.reshape(b, -1)
        yy = y.reshape(b, -1)
        return -0.5 * torch.sum((yh - yy) ** 2, 1).mean()
    if kind == 'gauss_var':                                     # train_particles.py:293-296,336
        yh = y_hat.reshape(b, -1)
        yy = y.reshape(b, -1)
        n = yy.shape[1]
        mu, logvar = yh[:, :n], yh[:, n:]
        return -0.5 * torch.sum((mu - yy) ** 2 / torch.exp(logvar) + logvar, 1).mean()
    raise ValueError(kind)


def get_latent(x_coord: Tensor, y: Tensor, enc: Dict[str, Tensor], R: int, padding: int, rot_refinement: bool,
               theta_prior: float = np.pi, normal_prior_over_r: bool = True):
    """clustering_mnist.py:121-161 (attention/attention branch): most probable (r,h,w) under attn, content
    (z_mu, exp(z_logstd)) and theta_mu there, dx = softmax-expected grid position summed over rotations."""
    b = y.shape[0]
    spacing = float((x_coord[1, 0] - x_coord[0, 0]).numpy())
    E = torch.ones(b, 1)       # the Gumbel sample is not used by get_latent
    Ho = y.shape[-1] + 2 * padding - enc['conv1.weight'].shape[-1] + 1
    attn, _, _, _, _, theta_vals, z_vals = encoder_forward(enc, y, E.expand(b, R * Ho * Ho), R, padding,
                                                            rot_refinement, theta_prior, normal_prior_over_r)
    _, ind1 = attn.reshape(b, -1).max(1)                                                # :127
    ind0 = torch.arange(b)
    z_vals = z_vals.reshape(b, z_vals.shape[1], -1)
    theta_vals = theta_vals.reshape(b, theta_vals.shape[1], -1)
    zd = z_vals.shape[1] // 2
    z_mu = z_vals[:, :zd][ind0, :, ind1]
    z_std = torch.exp(z_vals[:, zd:])[ind0, :, ind1]                                    # :137 (no epsilon)
    z_content = torch.cat((z_mu, z_std), dim=1)                                         # :142
    a_soft = F.softmax(attn.reshape(b, -1), dim=1).view(attn.shape).sum(1).view(b, -1).unsqueeze(2)   # :144
    G = torch.from_numpy(translation_grid(Ho, spacing))
    dx = torch.bmm(G.expand(b, G.shape[0], 2).transpose(1, 2).type(torch.float), a_soft).squeeze(2)   # :158
    theta_mu = theta_vals[ind0, 0:1, ind1]                                              # :161
    return z_content, theta_mu, dx


def particles_logp(y_hat: Tensor, y: Tensor, ctf: Optional[Tensor], mask_radius: int, dx: Tensor,
                   spacing: float) -> Tensor:
    """Particle likelihood tail, train_particles.py:284-338 (n_out = 1): optional per-image CTF filter
    (depthwise conv2d, :298-302), optional circular mask centred at the inferred translation (:309-333),
    Gaussian log-likelihood (:338).  ctf (B,1,kc,kc); dx (B,1,2)."""
    b = y.shape[0]
    n = int(y.shape[-1])
    y_mu = y_hat.reshape(b, -1)
    yy = y.reshape(b, -1)
    if ctf is not None:
        pad = ctf.size(2) // 2
        y_mu = F.conv2d(y_mu.view(1, -1, n, n), ctf, padding=pad, groups=ctf.size(0)).view(-1, n * n)
    if mask_radius > 0:
        x_img = np.arange(-n // 2, n // 2, 1)
        y_img = np.arange(n // 2, -n // 2, -1)
        xg, yg = np.meshgrid(x_img, y_img)
        grid = np.stack([xg.ravel(), yg.ravel()], 1)
        gb = np.broadcast_to(grid, (b, grid.shape[0], 2))
        center = dx.detach().numpy() / np.float32(spacing)
        dist = np.sqrt((center[:, :, 0] - gb[:, :, 0]) ** 2 + (center[:, :, 1] - gb[:, :, 1]) ** 2)
        mask = (torch.from_numpy(dist) < mask_radius).view(b, -1)
        yy = torch.where(mask, yy, torch.zeros_like(yy))
        y_mu = torch.where(mask, y_mu, torch.zeros_like(y_mu))
    return -0.5 * torch.sum((y_mu - yy) ** 2, 1).mean()


def ctf_filters(defocus, cs, voltage, apix, bfactor, ampcont, dfang, n, m, scale=1.0) -> np.ndarray:
    """Real-space CTF kernels, src/ctf.py:6-23,32-55 (one (n,m) float32 kernel per parameter row)."""
    theta, gamma = np.meshgrid(np.fft.fftfreq(n), np.fft.fftfreq(m), indexing='ij')
    freqs = np.stack([theta.ravel(), gamma.ravel()], 1)
    out = np.zeros((len(defocus), n, m), dtype=np.float32)
    for i in range(len(defocus)):
        f = freqs / (apix[i] * scale)
        volt = voltage[i] * 1000
        csv = cs[i] * 10 ** 7
        lam = 12.2639 / np.sqrt(volt + 0.97845e-6 * volt ** 2)
        x, yv = f[:, 0], f[:, 1]
        ang = np.arctan2(yv, x)
        s2 = x ** 2 + yv ** 2
        dfu = dfv = defocus[i] * 10000
        df = 0.5 * (dfu + dfv + (dfu - dfv) * np.cos(2 * (ang - 2 * np.pi * dfang[i] / 360)))
        gam = 2 * np.pi * (-0.5 * df * lam * s2 + 0.25 * csv * lam ** 3 * s2 ** 2)
        w = ampcont[i] / 100
        c = np.sqrt(1 - w ** 2) * np.sin(gam) - w * np.cos(gam)
        c = c * np.exp(-bfactor[i] / 4 * s2)
        out[i] = -np.fft.fftshift(np.fft.ifft2(c.reshape(n, m))).real
    return out


# --------------------------------------------------------------------------------------
# one optimiser step (train_mnist.py:311-324) -- used as the timed CPU baseline ("port")
# --------------------------------------------------------------------------------------
def adam_update(params, grads, m, v, step: int, lr: float = 2e-4, b1: float = 0.9,
                b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults (train_mnist.py:579), in place."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for p_, g_, m_, v_ in zip(params, grads, m, v):
        m_.mul_(b1).add_(g_, alpha=1 - b1)
        v_.mul_(b2).addcmul_(g_, g_, value=1 - b2)
        denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
        p_.addcdiv_(m_, denom, value=-lr / bc1)


def train_step(x_coord, y, enc, gen, opt_state, noise, **cfg):
    """fwd + bwd + Adam on dicts of leaf tensors (requires_grad=True).  Returns scalars."""
    names_e = sorted(enc)
    names_g = [k for k in sorted(gen) if not k.startswith('embed_latent')]
    leaves = [gen[k] for k in names_g] + [enc[k] for k in names_e]
    for t in leaves:
        t.grad = None
    elbo, log_p, kl = elbo_step(x_coord, y, enc, gen, E=noise['E'], eps_z=noise['eps_z'],
                                eps_theta=noise['eps_theta'], **cfg)
    (-elbo).backward()
    opt_state['step'] += 1
    with torch.no_grad():
        adam_update(leaves, [t.grad for t in leaves], opt_state['m'], opt_state['v'],
                    opt_state['step'], lr=opt_state.get('lr', 2e-4))
    return float(elbo), float(log_p), float(kl)


def new_opt_state(enc, gen, lr=2e-4):
    names_e = sorted(enc)
    names_g = [k for k in sorted(gen) if not k.startswith('embed_latent')]
    leaves = [gen[k] for k in names_g] + [enc[k] for k in names_e]
    return dict(step=0, lr=lr, m=[torch.zeros_like(t) for t in leaves],
                v=[torch.zeros_like(t) for t in leaves])
