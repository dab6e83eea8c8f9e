"""HIP step against the pinned CPU oracle AT THE BENCHMARK'S OWN SIZE (S64: 64x64, P8, z=2, k=64 p=16, C=128, hidden 512;
B = 64 and the full B = 256 of BASELINE.json's metric) -- every CU busy, persistent kernels with their full grids, the
split-K finalizers, slab sums and 8-XCD slice logic of the timed path, checked from OUTSIDE (the oracle restates
/root/reference/train_particles.py:186-343 and src/models.py:202-225,347-358,390-392; pinned by tests/golden).

Two tests:
* the whole ELBO step: ELBO / log p / KL within 1e-4, per-image kl, z, theta, dx within 1e-4, and every parameter gradient
  under the conditioning-aware gate max(1e-3, 2 x the ORACLE's own gradient change under a 1e-5 relative input perturbation);
* a KINK-FREE probe of the encoder (lifting convolution, conv2, heads -- forward, data gradients, weight gradients): the
  loss reads the head tensor only at positions where no LeakyReLU pre-activation of either layer lies within a margin of 0,
  so that no implementation-dependent kink flip can enter, and conv1.weight / conv2.weight / the head weights must then
  agree within 1e-3 of max-norm FLAT, no outlier rows, no conditioning allowance.
"""
import numpy as np
import pytest
import torch

from conftest import assert_grad_close, rel_err
from oracle import tvae_oracle as O

pytestmark = pytest.mark.gpu

S64 = dict(n=64, cin=1, zd=2, C=128, k=64, pad=16, R=8, hidden=512, layers=2, n_out=1, fourier=False, lik='gauss',
           data='randn')
# the other BASELINE configurations at THEIR stated sizes (round 4): MNIST shape 28x28 P8 (cfg2) and P16
# with the Fourier decoder (cfg3) at bs = 256, the galaxy shape 128x128x3 P16 z=50, four decoder layers, 3-channel BCE (cfg5)
# at B = 2 (its lifted activations are 136 MB per image; the oracle needs ~20 s for two)
CFGS = {
    'S64': S64,
    'S28': dict(n=28, cin=1, zd=2, C=128, k=28, pad=8, R=8, hidden=512, layers=2, n_out=1, fourier=False, lik='bce', data='rand'),
    'S28F': dict(n=28, cin=1, zd=2, C=128, k=28, pad=8, R=16, hidden=512, layers=2, n_out=1, fourier=True, lik='bce', data='rand'),
    'G128': dict(n=128, cin=3, zd=50, C=128, k=64, pad=32, R=16, hidden=512, layers=4, n_out=3, fourier=True, lik='bce3',
                 data='rand'),
}


def dev():
    return torch.device('cuda', 0)


def _models(scale_heads=10.0, seed=41, c=S64):
    import src.models as M
    torch.manual_seed(seed)                         # reference default init, generator first (train_mnist.py:522,551)
    gen = M.SpatialGenerator(c['zd'], c['hidden'], n_out=c['n_out'], num_layers=c['layers'],
                             fourier_expansion=c['fourier'], sigma=2.0 / (c['n'] - 1))
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        c['n'], c['cin'], c['zd'], kernels_num=c['C'], kernels_size=c['k'], padding=c['pad'], groupconv=c['R'],
        rot_refinement=True, theta_prior=np.pi, normal_prior_over_r=False)
    with torch.no_grad():
        for m in (enc.conv_a, enc.conv_r, enc.conv_z):
            m.weight.mul_(scale_heads)              # peaked attention: KL, theta, dx are O(1), not ~0 as at default init
    return gen, enc


def _inputs(B, seed=7, c=S64):
    g = torch.Generator().manual_seed(seed)
    ho = c['n'] + 2 * c['pad'] - c['k'] + 1
    mk = torch.randn if c['data'] == 'randn' else torch.rand
    y = mk(B, c['cin'], c['n'], c['n'], generator=g)
    E = torch.empty(B, c['R'] * ho * ho).exponential_(generator=g)
    return y, E, torch.randn(B, c['zd'], generator=g), torch.randn(B, generator=g)


def _oracle_step(y, enc_sd, gen_sd, E, ez, et, aux=False, c=S64):
    encp = {k_: v.detach().clone().requires_grad_(True) for k_, v in enc_sd.items()}
    genp = {k_: (v.detach().clone().requires_grad_(True) if not k_.startswith('embed_latent') else v.detach().clone())
            for k_, v in gen_sd.items()}            # (the random Fourier features are buffers: no gradient)
    out = O.elbo_step(O.image_coords(c['n']), y, encp, genp, R=c['R'], padding=c['pad'], rot_refinement=True,
                      theta_prior=np.pi, normal_prior_over_r=False, num_layers=c['layers'], likelihood=c['lik'],
                      fourier_sigma=(2.0 / (c['n'] - 1)) if c['fourier'] else None, E=E,
                      eps_z=ez, eps_theta=et, return_aux=aux)
    (-out[0]).backward()
    grads = {'e.' + k_: v.grad for k_, v in encp.items()}
    grads.update({'d.' + k_: v.grad for k_, v in genp.items() if v.requires_grad})
    return out, grads


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('cfg,B', [('S64', 64), ('S64', 256), ('S28', 256), ('S28F', 256), ('G128', 2)])
def test_step_matches_oracle_at_bench_size(cfg, B):
    from tvae import step
    c = CFGS[cfg]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    gen, enc = _models(c=c)
    y, E, ez, et = _inputs(B, c=c)
    enc_sd = {k_: v.clone() for k_, v in enc.state_dict().items()}
    gen_sd = {k_: v.clone() for k_, v in gen.state_dict().items()}
    (e_o, lp_o, kl_o, aux_o), g_o = _oracle_step(y, enc_sd, gen_sd, E, ez, et, aux=True, c=c)
    # the oracle's own conditioning at this size: gradient change under a 1e-5 relative perturbation of the images
    gper = torch.Generator().manual_seed(900)
    # (galaxy: 3e-5 -- the probe is sized to what it stands for, the rounding of the configuration's own convolution sums on
    # BOTH sides: sqrt(K) 2^-24 = 7e-6 of the output scale at K = 3 x 64 x 64 = 12 288 terms, 3.5x the K = 4 096 of S64, and
    # the kink-flip noise it causes in a gradient over 2 x 68 M pre-activations grows with it)
    eps_p = 3e-5 if cfg == 'G128' else 1e-5
    _, g_p = _oracle_step(y * (1.0 + eps_p * torch.randn(y.shape, generator=gper)), enc_sd, gen_sd, E, ez, et, c=c)
    cond = {k_: float((g_p[k_] - g_o[k_]).abs().max() / g_o[k_].abs().max().clamp_min(1e-30)) for k_ in g_o}
    del g_p
    if cfg == 'G128':
        # the galaxy step has a second, stronger sensitivity (see y_hat below): the pooled rotation / translation enter cos(.)
        # arguments with a derivative of ~250, so what two fp32 summation orders of the attention-weighted sums differ by
        # (~1e-6 of theta, dx) changes the reconstruction's gradient field.  Measured in the ORACLE: Gumbel noise times
        # 1 + 1e-6 N(0,1) (moves the sampled attention, hence theta / dx / z, by that much); the gate takes the larger of the two
        _, g_q = _oracle_step(y, enc_sd, gen_sd, E * (1.0 + 1e-6 * torch.randn(E.shape, generator=gper)), ez, et, c=c)
        for k_ in g_o:
            cond[k_] = max(cond[k_], float((g_q[k_] - g_o[k_]).abs().max() / g_o[k_].abs().max().clamp_min(1e-30)))
        del g_q

    gen, enc = gen.to(dev()), enc.to(dev())
    x = O.image_coords(c['n']).to(dev())
    noise = (E.to(dev()), ez.to(dev()), et.to(dev()))
    elbo, logp, kl, aux = step.elbo_terms(x, y.to(dev()), gen, enc, c['lik'], noise, return_aux=True)
    (-elbo).backward()
    torch.cuda.synchronize()
    for got, want, nm in ((elbo, e_o, 'elbo'), (logp, lp_o, 'log_p'), (kl, kl_o, 'kl')):
        assert abs(float(got) - float(want)) / abs(float(want)) < 1e-4, (cfg, nm, float(got), float(want))
    assert float(aux_o['a_sampled'].max()) > (0.05 if cfg != 'G128' else 1e-3)   # attention is peaked, not uniform (galaxy: 1 / 266 256)
    for k_ in ('kl_per_image', 'z', 'theta', 'dx'):
        assert rel_err(aux[k_].detach().reshape(-1), aux_o[k_].detach().reshape(-1)) < 1e-4, (cfg, k_)
    if cfg != 'G128':
        assert rel_err(aux['y_hat'].detach().reshape(-1), aux_o['y_hat'].detach().reshape(-1)) < 1e-4
    else:
        # The galaxy decoder evaluates cos(x' W / sigma + b) with sigma = 2 / 127: arguments of ~200 rad whose derivative with
        # respect to the pooled rotation / translation is ~250.  theta and dx are attention-weighted sums over 266 256
        # positions; they agree to ~1e-6 between the two implementations (different summation orders; asserted to 1e-4
        # above), which moves y_hat -- logits of size 0.04 -- by a few 1e-4 of its norm in ANY two fp32 evaluations.
        # The decoder itself is therefore checked on the SAME pooled sample: the oracle's generator, fed the HIP path's
        # (z, theta, dx) through the reference's coordinate transform (train_galaxy.py: x - dx, then the rotation),
        # must reproduce the HIP reconstruction to 1e-4; against the oracle's own sample the gate is 2e-3.
        with torch.no_grad():
            th, dxh, zh = aux['theta'].detach().cpu(), aux['dx'].detach().cpu().view(B, 1, 2), aux['z'].detach().cpu()
            xo = O.image_coords(c['n']).expand(B, c['n'] ** 2, 2) - dxh
            rot = torch.stack([torch.stack([torch.cos(th), torch.sin(th)], 1),
                               torch.stack([-torch.sin(th), torch.cos(th)], 1)], 1)
            y_same = O.generator_forward(gen_sd, torch.bmm(xo, rot).contiguous(), zh, c['layers'], False, 2.0 / (c['n'] - 1))
        assert rel_err(aux['y_hat'].detach().reshape(-1), y_same.reshape(-1)) < 1e-4
        assert rel_err(aux['y_hat'].detach().reshape(-1), aux_o['y_hat'].detach().reshape(-1)) < 2e-3
    gmax = max(float(v.abs().max()) for k_, v in g_o.items() if k_.startswith('e.'))
    for prefix, mod in (('e.', enc), ('d.', gen)):
        for k_, t in mod.named_parameters():
            want = g_o[prefix + k_]
            if prefix + k_ == 'e.conv_a.bias':       # analytically zero (softmax shift invariance): noise on both sides
                assert float(t.grad.abs().max()) <= 1e-5 * gmax and float(want.abs().max()) <= 1e-5 * gmax
                continue
            assert_grad_close(t.grad, want, tol=max(1e-3, 2 * cond[prefix + k_]),
                              floor=1e-3 * gmax if prefix == 'e.' else 0.0, name=f'{cfg} B={B} {prefix}{k_}')


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('cfg,B', [('S64', 64), ('S28F', 256), ('S28', 256), ('G128', 2)])
def test_decoder_kink_free_probe_at_bench_size(cfg, B):
    """The decoder's counterpart of the encoder probe below: SpatialGenerator at full width on B images,
    loss = sum(W * y_hat) with W = 0 on every pixel where ANY hidden pre-activation (512 per LeakyReLU layer) lies within
    MARGIN (relative to that layer's rms) of zero.  No kink flip can enter, so `layers.*.weight`, the coordinate / latent
    layers and every bias are held to 1e-3 of max-norm FLAT -- the recomputed first layer, the fused output dot, the
    two-valued implicit gradient with its sign bits and row sums, and the fused first-layer backward checked from outside.
    Reference: /root/reference/src/models.py:95-123."""
    import torch.nn.functional as F
    from tvae import ops
    MARGIN = 1e-4
    c = CFGS[cfg]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    gen, _ = _models(c=c)
    n, zd = c['n'], c['zd']
    g = torch.Generator().manual_seed(11)
    th = torch.rand(B, generator=g) * 2 * np.pi
    dx = 0.1 * torch.randn(B, 1, 2, generator=g)
    rot = torch.stack([torch.stack([th.cos(), -th.sin()], 1), torch.stack([th.sin(), th.cos()], 1)], 1)   # (B, 2, 2)
    xr = torch.bmm(O.image_coords(n).expand(B, n * n, 2) - dx, rot).contiguous()
    z = torch.randn(B, zd, generator=g)
    sig = 2.0 / (n - 1)
    gp = {k_: (v.detach().clone().requires_grad_(not k_.startswith('embed_latent'))) for k_, v in gen.state_dict().items()}
    # the oracle's forward, with the pre-activations kept (oracle/tvae_oracle.py:generator_forward restated line by line)
    h = xr.reshape(B * n * n, 2)
    if c['fourier']:
        h = torch.cos(F.linear(h, gp['embed_latent.weight'] / torch.tensor(sig), gp['embed_latent.bias']))
    pre = F.linear(h, gp['coord_linear.weight'], gp['coord_linear.bias']).view(B, n * n, -1) + \
        F.linear(z, gp['latent_linear.weight']).unsqueeze(1)
    pres = [pre.view(B * n * n, -1)]
    h = F.leaky_relu(pres[0])
    li = 1
    for _ in range(1, c['layers']):
        pres.append(F.linear(h, gp[f'layers.{li}.weight'], gp[f'layers.{li}.bias']))
        h = F.leaky_relu(pres[-1])
        li += 2
    yo = F.linear(h, gp[f'layers.{li}.weight'], gp[f'layers.{li}.bias']).view(B, n * n, -1)
    assert torch.equal(yo.detach(), O.generator_forward({k_: v.detach() for k_, v in gp.items()}, xr, z, c['layers'], False,
                                                        sig if c['fourier'] else None))
    with torch.no_grad():
        clean = torch.ones(B * n * n, dtype=torch.bool)
        for p_ in pres:
            clean &= (p_.abs() > MARGIN * p_.pow(2).mean().sqrt()).all(dim=1)
        frac = float(clean.float().mean())
        assert 0.5 < frac < 1.0, frac
        W = torch.randn(yo.shape, generator=g) * clean.view(B, n * n, 1)
    (yo * W).sum().backward()
    ops.PATH_LOG = set()
    try:
        gen = gen.to(dev())
        yh = gen(xr.to(dev()), z.to(dev()))
        (yh.view(B, n * n, -1) * W.to(dev())).sum().backward()
        torch.cuda.synchronize()
        took = set(ops.PATH_LOG)
    finally:
        ops.PATH_LOG = None
    # (round 5, VERDICT r04 weak #2: S28 @ 256 -- the padded pixel ranges take the S64 branches -- and the galaxy decoder)
    want_paths = ({'dec.virt_act', 'dec.fused_out', 'dec.virt_grad_2val', 'dec.sign_bits', 'dec.fuse_in',
                   'dec.row_sums_in_dgrad'} if cfg in ('S64', 'S28') else
                  {'dec.four_x6', 'dec.hidden_x6'} if cfg == 'G128' else {'dec.four_x6', 'dec.fused_out', 'dec.virt_grad_2val'})
    assert want_paths <= took, (want_paths - took, took)
    assert rel_err(yh.detach().reshape(-1), yo.detach().reshape(-1)) < 1e-4
    for k_, t in gen.named_parameters():
        want = gp[k_].grad
        err = float((t.grad.detach().cpu().double() - want.double()).abs().max() / want.double().abs().max())
        assert err < 1e-3, (cfg, f'B={B}', k_, err)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('cfg,B', [('S64', 64), ('S64', 256), ('G128', 2), ('S28', 256), ('S28F', 256)])
def test_encoder_kink_free_probe_at_bench_size(cfg, B):
    """conv1 -> LeakyReLU -> conv2 -> LeakyReLU -> heads at full size, loss = sum(W * heads) with W = 0 on every position
    (b, r, h, w) where ANY of the 128 + 128 pre-activations is within MARGIN (relative to that layer's rms) of zero.  A kink
    flip needs a pre-activation that two fp32 implementations place on different sides of 0, i.e. |pre| of the order of
    1e-6 x scale: none survives a 1e-4 margin, so the comparison is kink free and is held to 1e-3 FLAT."""
    from tvae import ops
    MARGIN = 1e-4
    torch.set_num_threads(min(16, torch.get_num_threads()))
    c = CFGS[cfg]                                    # (G128: the 3-channel / 192-wide-frame convolution with its generic transforms)
    _, enc = _models(c=c)
    y, _, _, _ = _inputs(B, seed=8, c=c)
    encp = {k_: v.detach().clone().requires_grad_(True) for k_, v in enc.state_dict().items()}
    heads_o, pre1, pre2 = O.encoder_heads(encp, y, c['R'], c['pad'])          # (B, 7, R, Ho, Ho), (B, C, R, Ho, Ho) x 2
    with torch.no_grad():
        clean = ((pre1.abs() > MARGIN * pre1.pow(2).mean().sqrt()).all(dim=1) &
                 (pre2.abs() > MARGIN * pre2.pow(2).mean().sqrt()).all(dim=1))                 # (B, R, Ho, Ho)
        frac = float(clean.float().mean())
        assert 0.5 < frac < 1.0, frac                  # most positions are usable, and the mask is not vacuous
        g = torch.Generator().manual_seed(3)
        W = torch.randn(heads_o.shape, generator=g) * clean.unsqueeze(1)
    del pre1, pre2
    (heads_o * W).sum().backward()
    g_o = {k_: v.grad for k_, v in encp.items()}

    enc = enc.to(dev())
    ops.PATH_LOG, ops.PARTS_LOG = set(), {}
    try:
        heads = enc.encode_heads(y.to(dev()))                                  # [7][B*R*Ho*Ho] feature-major
        Wd = W.permute(1, 0, 2, 3, 4).reshape(heads.shape[0], -1).contiguous().to(dev())
        (heads * Wd).sum().backward()
        torch.cuda.synchronize()
        took, parts_log = set(ops.PATH_LOG), dict(ops.PARTS_LOG)
    finally:
        ops.PATH_LOG, ops.PARTS_LOG = None, None
    # the timed branches ran (galaxy: 103 head rows -- the encoder tail takes the unfused fp32-MFMA layers)
    tail = {'enc.tail_fwd_x6', 'enc.tail_dgrad_x6', 'enc.tail_wgrad_x6'} if cfg != 'G128' else set()
    assert ({'conv1.dft'} | tail | ({'conv1.dft_ring'} if cfg != 'G128' else set())) <= took, took
    from tvae import _lib
    if _lib.get_gemm_mode() == 'h3':                     # ... in their two-part instances (parts as passed to the C ABI)
        for k_ in ('tvae_conv1_fwd', 'tvae_conv1_wgrad') + (('tvae_enc_tail_fwd_x6', 'tvae_enc_tail_dgrad_x6',
                                                             'tvae_enc_tail_wgrad_x6') if tail else ()):
            assert parts_log.get(k_) == [(2, 3)], (k_, parts_log.get(k_))
    ho = heads_o.shape[-1]
    want_h = heads_o.detach().permute(1, 0, 2, 3, 4).reshape(heads.shape[0], -1)
    assert rel_err(heads.detach(), want_h) < 1e-4
    for k_, t in enc.named_parameters():
        err = float((t.grad.detach().cpu().double() - g_o[k_].double()).abs().max() / g_o[k_].double().abs().max())
        assert err < 1e-3, (cfg, f'B={B}', k_, err)


def test_inference_forward_at_bench_size_skips_training_stores():
    """S64, B = 256 (every CU busy, full persistent grids): the no-grad forward (eval_model / get_latent path) is bitwise the
    training forward, allocates >= 1.1 GB less (the conv2 activation H [128][2.2 M] fp32 and the two sign-word planes are
    never created) and leaves nothing behind on the module."""
    from tvae import ops, step
    c = S64
    B = 256
    gen, enc = _models(c=c)
    gen, enc = gen.to(dev()), enc.to(dev())
    y, E, ez, et = (t.to(dev()) for t in _inputs(B, c=c))
    x = O.image_coords(c['n']).to(dev())
    for _ in range(2):                                   # scratch buffers and tables exist afterwards
        with torch.no_grad():
            step.elbo_terms(x, y, gen, enc, c['lik'], (E, ez, et))
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        ei = step.elbo_terms(x, y, gen, enc, c['lik'], (E, ez, et), return_aux=True)
    torch.cuda.synchronize()
    peak_inf = torch.cuda.max_memory_allocated() - base
    hi, yi = ei[3]['heads'].clone(), ei[3]['y_hat'].clone()
    ei = tuple(t.clone() for t in ei[:3])
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    ops.PATH_LOG = set()
    try:
        etr = step.elbo_terms(x, y, gen, enc, c['lik'], (E, ez, et), return_aux=True)
        took = set(ops.PATH_LOG)
    finally:
        ops.PATH_LOG = None
    torch.cuda.synchronize()
    peak_tr = torch.cuda.max_memory_allocated() - base
    assert 'enc.inference' not in took and 'dec.no_h' in took
    for a, b in zip(ei, etr[:3]):
        assert torch.equal(a, b.detach())
    assert torch.equal(hi, etr[3]['heads'].detach()) and torch.equal(yi, etr[3]['y_hat'].detach())
    ho = c['n'] + 2 * c['pad'] - c['k'] + 1
    h_bytes = 4 * c['C'] * B * c['R'] * ho * ho
    assert peak_tr - peak_inf >= h_bytes, (peak_tr, peak_inf, h_bytes)


# bf16 throughput mode (TVAE_GEMM=bf16; BASELINE.json configs[1] and [4] name it) AT THE STATED SIZES against the fp32 oracle.
# It is not fp32-equivalent (operands of every split-pipe product rounded to ONE bf16 number, T / S' of the convolution stored
# as bf16), so it has its own gate, stated here and measured on the driver's GPU by this test (the measured values are written
# to gpurun_out/bf16_bench_size_<cfg>.json): per quantity the bound is ~2x what was measured when the gate was set (round 5).
BF16_GATE = {
    # measured (round 5, MI355X): S28 @ 256: ELBO terms 3e-5 / 3e-5 / 8e-5, z / theta / dx 3.6e-3 / 1.2e-3 / 8e-4, y_hat 7e-3,
    #   every gradient tensor <= 5.7e-3 relative L2 at cosine >= 0.99999;
    # S64 @ 64: ELBO 4e-5, latents <= 2.9e-3, y_hat 5.5e-3, gradients <= 6.2e-3 except conv1.weight 5.5e-2 (cosine 0.9985: T and S'
    #   stored as bf16 on top of one-part operands);
    # G128 @ 2: ELBO terms <= 3e-5, latents <= 6e-4, y_hat 1.5e-2 -- but the ENCODER gradients only to 0.33-0.36 relative L2 at cosine
    #   0.945: at the galaxy shape with two images the gradient is ill-conditioned in ANY arithmetic (a 1e-6 relative input
    #   perturbation moves the fp32 oracle's own gradients by 2-4e-3, test_step_matches_oracle_at_bench_size), and bf16's 4e-3
    #   operand rounding is amplified by the same factor.  The gate there only holds the mode to "same direction"; DESIGN.md says so.
    'S28': dict(elbo=2e-4, latent=1e-2, yhat=2e-2, grad_l2=1.5e-2, cos=0.9999),
    'S64': dict(elbo=2e-4, latent=1e-2, yhat=2e-2, grad_l2=1.2e-1, cos=0.997),
    'G128': dict(elbo=2e-4, latent=5e-3, yhat=4e-2, grad_l2=6e-1, cos=0.9),
}


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('cfg,B', [('S28', 256), ('S64', 64), ('G128', 2)])
def test_bf16_mode_matches_oracle_at_bench_size(cfg, B):
    import json
    import os
    from tvae import _lib, step
    c = CFGS[cfg]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    gen, enc = _models(c=c)
    y, E, ez, et = _inputs(B, c=c)
    enc_sd = {k_: v.clone() for k_, v in enc.state_dict().items()}
    gen_sd = {k_: v.clone() for k_, v in gen.state_dict().items()}
    (e_o, lp_o, kl_o, aux_o), g_o = _oracle_step(y, enc_sd, gen_sd, E, ez, et, aux=True, c=c)
    gen, enc = gen.to(dev()), enc.to(dev())
    x = O.image_coords(c['n']).to(dev())
    noise = (E.to(dev()), ez.to(dev()), et.to(dev()))
    with _lib.arithmetic('bf16'):
        elbo, logp, kl, aux = step.elbo_terms(x, y.to(dev()), gen, enc, c['lik'], noise, return_aux=True)
    (-elbo).backward()                                   # (runs in its forward's arithmetic)
    torch.cuda.synchronize()
    gate = BF16_GATE[cfg]
    rep = {'cfg': cfg, 'B': B,
           'elbo': abs(float(elbo) - float(e_o)) / abs(float(e_o)), 'log_p': abs(float(logp) - float(lp_o)) / abs(float(lp_o)),
           'kl': abs(float(kl) - float(kl_o)) / abs(float(kl_o))}
    for k_ in ('z', 'theta', 'dx', 'y_hat'):
        rep[k_] = rel_err(aux[k_].detach().reshape(-1), aux_o[k_].detach().reshape(-1))
    grads = {}
    for prefix, mod in (('e.', enc), ('d.', gen)):
        for k_, t in mod.named_parameters():
            if prefix + k_ == 'e.conv_a.bias':
                continue                                 # analytically zero
            a, b = t.grad.double().cpu().reshape(-1), g_o[prefix + k_].double().reshape(-1)
            grads[prefix + k_] = {'l2': float((a - b).norm() / b.norm()), 'cos': float(torch.dot(a, b) / (a.norm() * b.norm())),
                                  'max': float((a - b).abs().max() / b.abs().max())}
    rep['grads'] = grads
    rep['worst_grad_l2'] = max(v['l2'] for v in grads.values())
    rep['worst_grad_cos'] = min(v['cos'] for v in grads.values())
    try:
        os.makedirs('gpurun_out', exist_ok=True)
        json.dump(rep, open(os.path.join('gpurun_out', f'bf16_bench_size_{cfg}.json'), 'w'), indent=1)
    except OSError:
        pass
    for k_ in ('elbo', 'log_p', 'kl'):
        assert rep[k_] < gate['elbo'], (cfg, k_, rep[k_])
    for k_ in ('z', 'theta', 'dx'):
        assert rep[k_] < gate['latent'], (cfg, k_, rep[k_])
    assert rep['y_hat'] < gate['yhat'], (cfg, rep['y_hat'])
    for k_, v in grads.items():
        assert v['l2'] < gate['grad_l2'] and v['cos'] > gate['cos'], (cfg, k_, v)
