"""ELBO step and epoch loops of TARGET-VAE on the HIP kernels.

`eval_minibatch`, `train_epoch` and `eval_model` keep the reference signatures and return values
(train_mnist.py:26-27, :300-301, :352-353; the particles variant train_particles.py:28-29 adds `ctf`
and replaces `image_dim` by `padding, mask_radius`).  Only the TARGET-VAE configuration
(t_inf = attention, r_inf = attention | attention+offsets) is on the hand-written path.
"""
from __future__ import annotations

import sys
from typing import Optional

import torch

from . import ops

_SPACING = {}


def pixel_spacing(x_coord: torch.Tensor) -> float:
    """float32 pixel spacing x[1,0]-x[0,0] (reference reads it from the device EVERY step,
    train_mnist.py:30; it only depends on the image size, so it is cached per coordinate tensor)."""
    key = (x_coord.data_ptr(), tuple(x_coord.shape), str(x_coord.device))
    v = _SPACING.get(key)
    if v is None:
        v = float((x_coord[1, 0] - x_coord[0, 0]).item())
        _SPACING[key] = v
    return v


_NOISE_GEN = {}


def set_noise_generator(device, generator: Optional[torch.Generator]) -> None:
    """Generator the per-step draws of `elbo_terms` use on `device` (None = torch's default device generator).
    Data-parallel runs give every rank its own stream so that the noise of a global minibatch is independent across
    its shards, as it is in a single-process run (tvae/driver.py)."""
    _NOISE_GEN[(torch.device(device).type, torch.device(device).index)] = generator


def draw_noise(B: int, RP: int, zd: int, device, generator: Optional[torch.Generator] = None):
    """The three per-step draws of the reference, in its order: Exp(1) for the Gumbel-softmax
    (models.py:387), N(0,1) for z (train_mnist.py:206) and for theta (train_mnist.py:230)."""
    if generator is None and _NOISE_GEN:
        d = torch.device(device)
        generator = _NOISE_GEN.get((d.type, d.index))
    E = torch.empty(B, RP, dtype=torch.float32, device=device).exponential_(generator=generator)
    eps_z = torch.randn(B, zd, dtype=torch.float32, device=device, generator=generator)
    eps_t = torch.randn(B, dtype=torch.float32, device=device, generator=generator)
    return E, eps_z, eps_t


def elbo_terms(x, y, generator_model, encoder_model, likelihood='bce', noise=None, return_aux=False, ctf=None,
               mask_radius=0):
    """(elbo f64, log_p f32, kl f64) of one minibatch, attention/attention branch
    (train_mnist.py:187-294).  `noise` = (E, eps_z, eps_theta) injects the random draws."""
    b = y.shape[0]
    dev = y.device
    enc = encoder_model
    R, Ho, zd = enc.groupconv, enc.output_size(), enc.latent_dim
    heads = enc.encode_heads(y)
    tb = enc.head_tables(dev, pixel_spacing(x))
    if noise is None:
        noise = draw_noise(b, R * Ho * Ho, zd, dev)
    E, eps_z, eps_t = noise
    attn, q, a, z, theta, dx, kl_b = ops.HeadFn.apply(heads, E.reshape(b, -1), eps_z.reshape(b, zd),
                                                      eps_t.reshape(b), tb, b, zd)
    xr = ops.CoordFn.apply(x, dx, theta)
    y_hat = generator_model(xr, z)
    if ctf is not None or mask_radius > 0:
        # particle tail (train_particles.py:298-338): CTF filter, then circular mask, then Gaussian likelihood
        if likelihood != 'gauss':
            raise NotImplementedError('CTF / mask with --fit-noise is inconsistent in the reference '
                                      '(train_particles.py:303-307,330-333 do not broadcast); only n_out = 1 is built')
        n = int(y.shape[-1])
        y_mu = y_hat.reshape(b, -1)
        if ctf is not None:
            y_mu = ops.CtfFn.apply(y_mu, ctf, n)
        if mask_radius > 0:
            lp = ops.MaskedLogLikFn.apply(y_mu, y, dx, pixel_spacing(x), mask_radius, n)
        else:
            lp = ops.LogLikFn.apply(y_mu, y, ops.LIK_KIND['gauss'])
    else:
        lp = ops.LogLikFn.apply(y_hat, y, ops.LIK_KIND[likelihood])
    # log_p = lp.mean() (float32: train_mnist.py:291), kl_div = kl_b.double().mean() (float64 like the reference, whose prior
    # grid is float64), elbo = log_p - kl_div -- one launch (ops.ElboFn)
    elbo, log_p, kl_div = ops.ElboFn.apply(lp, kl_b)
    if return_aux:
        return elbo, log_p, kl_div, dict(attn=attn, q_t_r=q, a_sampled=a, z=z, theta=theta, dx=dx, x_rot=xr,
                                         y_hat=y_hat, kl_per_image=kl_b, heads=heads)
    return elbo, log_p, kl_div


_NEG_ONE = {}


def backward_neg_elbo(elbo: torch.Tensor) -> None:
    """`(-elbo).backward()` of the reference's training loop (train_mnist.py:320-321: loss = -elbo; loss.backward()) without
    the negation, its backward and the ones-fill: the seed gradient d(-elbo)/d(elbo) = -1 is a cached constant."""
    key = (elbo.device.type, elbo.device.index, elbo.dtype)
    g = _NEG_ONE.get(key)
    if g is None:
        g = torch.full((), -1.0, dtype=elbo.dtype, device=elbo.device)
        _NEG_ONE[key] = g
    elbo.backward(gradient=g)


def _check_branch(t_inf, r_inf):
    if not (t_inf == 'attention' and r_inf in ('attention', 'attention+offsets')):
        raise NotImplementedError(
            f'--t-inf {t_inf} --r-inf {r_inf}: only the TARGET-VAE attention/attention(+offsets) branch '
            '(train_mnist.py:187-282) runs on the HIP hot path')


def _secondary(x, y, generator_model, encoder_model, t_inf, r_inf, theta_prior, likelihood, noise):
    """The two secondary branches (train_mnist.py:35-185) on the generic path of tvae/secondary.py."""
    from . import secondary
    if t_inf == 'unimodal' and r_inf == 'unimodal':
        return secondary.unimodal_unimodal(x, y, generator_model, encoder_model, theta_prior, likelihood, eps=noise)
    if t_inf == 'attention' and r_inf == 'unimodal':
        return secondary.attention_unimodal(x, y, generator_model, encoder_model, theta_prior, pixel_spacing(x),
                                            likelihood, noise=noise)
    raise NotImplementedError(f'--t-inf {t_inf} --r-inf {r_inf} is not a combination the reference supports '
                              '(train_mnist.py:35,86,187)')


def eval_minibatch(x, y, generator_model, encoder_model, t_inf, r_inf, epoch, device, theta_prior, groupconv,
                   image_dim, likelihood='bce', noise=None):
    """Reference signature train_mnist.py:26-27 (+ optional `likelihood`, `noise` keywords)."""
    if not (t_inf == 'attention' and r_inf in ('attention', 'attention+offsets')):
        return _secondary(x.to(device), y.to(device), generator_model, encoder_model, t_inf, r_inf, theta_prior,
                          likelihood, noise)
    return elbo_terms(x.to(device), y.to(device), generator_model, encoder_model, likelihood, noise)


def eval_minibatch_particles(x, y, ctf, generator_model, encoder_model, t_inf, r_inf, epoch, device, theta_prior,
                             groupconv, padding, mask_radius, noise=None):
    """Reference signature train_particles.py:28-29 (ctf: (B,1,kc,kc) filters of this minibatch or None)."""
    _check_branch(t_inf, r_inf)
    n_out = list(generator_model.layers)[-1].out_features
    return elbo_terms(x.to(device), y.to(device), generator_model, encoder_model,
                      'gauss_var' if n_out == 2 else 'gauss', noise, ctf=None if ctf is None else ctf.to(device),
                      mask_radius=mask_radius)


def train_epoch(iterator, x_coord, generator_model, encoder_model, optim, t_inf, r_inf, epoch, num_epochs, N, device,
                params, theta_prior, groupconv, image_dim, likelihood='bce', progress=True, noise_iter=None,
                mask_radius=None, graphed=None):
    """Reference train_mnist.py:300-346: loss = -elbo; backward; step; batch-weighted running means.
    With `mask_radius` not None it is the particles variant (train_particles.py:350-410): minibatches are (y,) or
    (y, ctf) and `image_dim` carries the encoder padding like the reference's positional argument.
    `graphed` (tvae.graph.GraphedStep, opt-in): minibatches of its captured size replay a hipGraph of forward + backward."""
    generator_model.train()
    encoder_model.train()
    c = 0
    gen_loss_accum = kl_loss_accum = elbo_accum = 0.0
    for mb in iterator:
        y = mb[0]
        b = y.size(0)
        noise = next(noise_iter) if noise_iter is not None else None
        if b == 0:
            # ragged tail smaller than the number of ranks: this rank has no image of the global minibatch, but it
            # still joins the gradient all-reduce (zero gradient, reducer weight 0) and takes the same Adam step
            optim.step()
            optim.zero_grad()
            continue
        if graphed is not None and b == graphed.B and len(mb) == 1 and mask_radius in (None, 0):
            # captured forward + backward (tvae/graph.py): a replay instead of ~70 launches; bitwise the eager result
            stats_t = graphed.run(y, noise)
            optim.step()
            stats = stats_t.tolist()
            optim.zero_grad()
        else:
            if mask_radius is not None:
                elbo, log_p, kl = eval_minibatch_particles(x_coord, y, mb[1] if len(mb) > 1 else None, generator_model,
                                                           encoder_model, t_inf, r_inf, epoch, device, theta_prior,
                                                           groupconv, image_dim, mask_radius, noise)
            else:
                elbo, log_p, kl = eval_minibatch(x_coord, y, generator_model, encoder_model, t_inf, r_inf, epoch, device,
                                                 theta_prior, groupconv, image_dim, likelihood, noise)
            backward_neg_elbo(elbo)
            optim.step()
            optim.zero_grad(set_to_none=True)
            stats = torch.stack([elbo.detach().double(), log_p.detach().double(), kl.detach().double()]).tolist()
        elbo_v, gen_loss, kl_loss = stats[0], -stats[1], stats[2]   # one sync instead of three .item()
        c += b
        gen_loss_accum += b * (gen_loss - gen_loss_accum) / c
        elbo_accum += b * (elbo_v - elbo_accum) / c
        kl_loss_accum += b * (kl_loss - kl_loss_accum) / c
        if progress:
            line = '# [{}/{}] training {:.1%}, ELBO={:.5f}, Error={:.5f}, KL={:.5f}'.format(
                epoch + 1, num_epochs, c / N, elbo_accum, gen_loss_accum, kl_loss_accum)
            print(line, end='\r', file=sys.stderr)
    if progress:
        print(' ' * 150, end='\r', file=sys.stderr)
    return elbo_accum, gen_loss_accum, kl_loss_accum


def eval_model(iterator, x_coord, generator_model, encoder_model, t_inf, r_inf, epoch, device, theta_prior,
               groupconv, image_dim, likelihood='bce', mask_radius=None, noise_iter=None):
    """Reference train_mnist.py:352-387 (noise is still drawn in eval); particles variant
    (train_particles.py:413-470) when `mask_radius` is not None."""
    generator_model.eval()
    encoder_model.eval()
    c = 0
    gen_loss_accum = kl_loss_accum = elbo_accum = 0.0
    with torch.no_grad():
        for mb in iterator:
            y = mb[0]
            b = y.size(0)
            noise = next(noise_iter) if noise_iter is not None else None
            if b == 0:
                continue                     # empty shard of a ragged tail: nothing to evaluate on this rank
            if mask_radius is not None:
                elbo, log_p, kl = eval_minibatch_particles(x_coord, y, mb[1] if len(mb) > 1 else None,
                                                           generator_model, encoder_model, t_inf, r_inf, epoch, device,
                                                           theta_prior, groupconv, image_dim, mask_radius, noise)
            else:
                elbo, log_p, kl = eval_minibatch(x_coord, y, generator_model, encoder_model, t_inf, r_inf, epoch,
                                                 device, theta_prior, groupconv, image_dim, likelihood, noise)
            stats = torch.stack([elbo.double(), log_p.double(), kl.double()]).tolist()
            c += b
            gen_loss_accum += b * (-stats[1] - gen_loss_accum) / c
            elbo_accum += b * (stats[0] - elbo_accum) / c
            kl_loss_accum += b * (stats[2] - kl_loss_accum) / c
    return elbo_accum, gen_loss_accum, kl_loss_accum
