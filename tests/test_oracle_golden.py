"""Pin the CPU oracle (oracle/tvae_oracle.py) against fixtures generated from the real reference
(tests/golden/make_goldens.py).  CPU-only; runs under `-m "not gpu"`."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_grad_close, assert_trajectory_update_close, load_golden, rel_err, tdict
from oracle import tvae_oracle as O

TOL = 2e-5   # oracle restates the same ATen CPU ops; only summation order differs


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'bank_*.npz'))))
def test_bank(path):
    fx = load_golden(os.path.basename(path)[:-4])
    w = torch.from_numpy(fx['weight']).requires_grad_(True)
    R = fx['bank'].shape[1]
    tw = O.rotated_bank(w, R)
    assert rel_err(tw, fx["bank"]) < 1e-5
    (tw * torch.from_numpy(fx['gbank'])).sum().backward()
    assert rel_err(w.grad, fx["gweight"]) < 1e-5


def test_bank_known_answers():
    """r=0 is the identity and r=R/4 is rot90(w,-1) (property ii)."""
    torch.manual_seed(0)
    w = torch.randn(2, 1, 1, 12, 12)
    tw = O.rotated_bank(w, 8)
    assert rel_err(tw[:, 0], w) < 1e-6
    assert rel_err(tw[:, 2], torch.rot90(w, -1, dims=(3, 4))) < 1e-6


@pytest.mark.parametrize('name', ['groupconv_fwd_bwd', 'groupconv_cin3_k9_R4'])
def test_groupconv(name):
    fx = load_golden(name)
    w = torch.from_numpy(fx['weight']).requires_grad_(True)
    b = torch.from_numpy(fx['bias']).requires_grad_(True)
    R = fx['out'].shape[2]
    n, k, Ho = fx['y'].shape[-1], fx['weight'].shape[-1], fx['out'].shape[-1]
    pad = (Ho - 1 + k - n) // 2
    out = O.groupconv_forward(torch.from_numpy(fx['y']), w, b, R, pad)
    assert rel_err(out, fx['out']) < TOL
    (out * torch.from_numpy(fx['gout'])).sum().backward()
    assert rel_err(w.grad, fx['gweight']) < TOL
    assert rel_err(b.grad, fx['gbias']) < TOL


def test_groupconv_rot90_equivariance():
    """gc(rot90(y)) == rot90(roll(gc(y), -R/4, dim=2)) for even k, symmetric pad (i)."""
    torch.manual_seed(1)
    R = 8
    w = torch.randn(3, 1, 1, 12, 12) * 0.1
    y = torch.rand(2, 1, 12, 12)
    a = O.groupconv_forward(torch.rot90(y, 1, dims=(2, 3)), w, None, R, 4)
    b = torch.rot90(torch.roll(O.groupconv_forward(y, w, None, R, 4), -R // 4, dims=2), 1, dims=(3, 4))
    assert rel_err(a, b) < 1e-5


@pytest.mark.parametrize('name', ['encoder_P8_28', 'encoder_P16_28_normal', 'encoder_P4_20_norefine',
                                  'encoder_P8_64'])
def test_encoder(name):
    fx = load_golden(name)
    n, cin, zd, C, k, p, R, refine, normal = [int(v) for v in fx['cfg']]
    prm = tdict(fx, 'p.', requires_grad=True)
    outs = O.encoder_forward(prm, torch.from_numpy(fx['y']), torch.from_numpy(fx['E']), R, p,
                             bool(refine), float(fx['theta_prior']), bool(normal))
    attn, q, p_r, a_s, offs, theta, z = outs
    for got, key in ((attn, 'attn'), (q, 'q_t_r'), (p_r, 'p_r'), (a_s, 'a_sampled'), (offs, 'offsets'),
                     (theta, 'theta'), (z, 'z')):
        assert rel_err(got, fx[key]) < TOL, key
    assert abs(float(torch.exp(q).reshape(q.shape[0], -1).sum(1).max()) - 1) < 1e-5
    # O.encoder_heads (the head stack + pre-activations the kink-free bench-size probe uses) against the same fixture:
    # theta / z are its rows 1:3 / 3: (before the offsets are added), attn its row 0 plus the prior
    hd, pre1, pre2 = O.encoder_heads(prm, torch.from_numpy(fx['y']), R, p)
    off = torch.from_numpy(fx['offsets']).view(1, R, 1, 1)
    assert rel_err(hd[:, 0] + torch.from_numpy(fx['p_r']).view(1, R, 1, 1), fx['attn']) < TOL
    assert rel_err(hd[:, 1] + off, fx['theta'][:, 0]) < TOL and rel_err(hd[:, 2], fx['theta'][:, 1]) < TOL
    assert rel_err(hd[:, 3:], fx['z']) < TOL and pre1.shape == pre2.shape == (hd.shape[0], C, R) + hd.shape[3:]
    w = {k_: torch.from_numpy(fx[k_]) for k_ in ('w_q', 'w_a', 'w_t', 'w_z')}
    probe = (q * w['w_q']).sum() + (a_s * w['w_a']).sum() * 50 + (theta * w['w_t']).sum() \
        + (z * w['w_z']).sum() + (attn * w['w_q']).sum() * 0.5
    probe.backward()
    floor = 1e-3 * max(float(np.abs(fx['g.' + k_]).max()) for k_ in prm)
    for k_, t in prm.items():
        assert_grad_close(t.grad, fx['g.' + k_], floor=floor, name=k_)


@pytest.mark.parametrize('name', ['decoder_plain', 'decoder_plain512', 'decoder_fourier', 'decoder_resid',
                                  'decoder_nout2', 'decoder_nout3_z50_L4', 'decoder_z0_L1'])
def test_decoder(name):
    fx = load_golden(name)
    zd, hid, n_out, L, resid, fourier = [int(v) for v in fx['cfg']]
    prm = tdict(fx, 'p.', requires_grad=True)
    x = torch.from_numpy(fx['x']).requires_grad_(True)
    z = torch.from_numpy(fx['z']).requires_grad_(True) if zd > 0 else None
    yh = O.generator_forward(prm, x, z, L, bool(resid), float(fx['sigma']) if fourier else None)
    assert rel_err(yh, fx['y_hat']) < TOL
    (yh * torch.from_numpy(fx['gy'])).sum().backward()
    assert_grad_close(x.grad.reshape(-1, 2), fx['gx'].reshape(-1, 2), name='gx')
    if zd > 0:
        assert_grad_close(z.grad, fx['gz'], name='gz')
    for k_, t in prm.items():
        if ('g.' + k_) in fx:
            assert_grad_close(t.grad, fx['g.' + k_], name=k_)


STEP_LIK = {'step_mnist28_P8_init': 'bce', 'step_mnist28_P8_peaked': 'bce',
            'step_mnist28_P16_fourier_normal': 'bce', 'step_mnist28_P4_attention_resid': 'bce',
            'step_particles64_P8': 'gauss', 'step_particles32_fitnoise': 'gauss_var',
            'step_galaxy_small': 'bce3'}


def step_cfg(fx):
    n, cin, zd, C, k, p, R, refine, normal, hid, L, n_out, fourier, resid = [int(v) for v in fx['cfg']]
    return dict(R=R, padding=p, rot_refinement=bool(refine), theta_prior=float(fx['theta_prior']),
                normal_prior_over_r=bool(normal), num_layers=L, resid=bool(resid),
                fourier_sigma=float(fx['sigma']) if fourier else None), n


@pytest.mark.parametrize('name', sorted(STEP_LIK))
def test_step(name):
    fx = load_golden(name)
    cfg, n = step_cfg(fx)
    enc = tdict(fx, 'e.', requires_grad=True)
    gen = tdict(fx, 'd.', requires_grad=True)
    elbo, logp, kl = O.elbo_step(O.image_coords(n), torch.from_numpy(fx['y']), enc, gen,
                                 likelihood=STEP_LIK[name], E=torch.from_numpy(fx['E']),
                                 eps_z=torch.from_numpy(fx['eps_z']),
                                 eps_theta=torch.from_numpy(fx['eps_theta']), **cfg)
    assert elbo.dtype == torch.float64 and kl.dtype == torch.float64 and logp.dtype == torch.float32
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < 1e-6
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < 1e-6
    assert abs(float(kl) - float(fx['kl'])) / abs(float(fx['kl'])) < 1e-5
    (-elbo).backward()
    floor = 1e-3 * max(float(np.abs(v).max()) for k_, v in fx.items() if k_.startswith('ge.'))
    for k_, t in enc.items():
        assert_grad_close(t.grad, fx['ge.' + k_], floor=floor, name=k_)
    for k_, t in gen.items():
        if ('gd.' + k_) in fx:
            assert_grad_close(t.grad, fx['gd.' + k_], name=k_)


@pytest.mark.parametrize('name,lik', [('hot_S64_B2', 'gauss'), ('hot_S28F_B8', 'bce'), ('hot_M50_B2', 'bce')])
def test_step_hot_widths(name, lik):
    """Full-width steps (hidden 512, C=128; S64 and S28F shapes) from the real reference: the oracle is pinned at the
    widths the benchmark runs, and the seeded drop-in construction reproduces the reference's parameters."""
    from conftest import seeded_models
    fx = load_golden(name)
    cfg, n = step_cfg(fx)
    enc_m, gen_m, _ = seeded_models(fx)
    enc = {k_: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k_, v in enc_m.state_dict().items()}
    gen = {k_: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k_, v in gen_m.state_dict().items()}
    elbo, logp, kl = O.elbo_step(O.image_coords(n), torch.from_numpy(fx['y']), enc, gen, likelihood=lik,
                                 E=torch.from_numpy(fx['E']), eps_z=torch.from_numpy(fx['eps_z']),
                                 eps_theta=torch.from_numpy(fx['eps_theta']), **cfg)
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < 1e-6
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < 1e-6
    assert abs(float(kl) - float(fx['kl'])) / abs(float(fx['kl'])) < 1e-5
    (-elbo).backward()
    # two fp32 CPU evaluations with different summation orders: 5e-4 of max-norm (half the 1e-3 parity gate of
    # SURVEY 8d) -- the fixtures' own kink conditioning (`ke.*` / `kd.*`: gradient change of the reference under a 1e-5
    # input perturbation) is 1e-4 .. 8e-3 at these widths; no free outlier for single-row tensors
    floor = 1e-3 * max(float(np.abs(v).max()) for k_, v in fx.items() if k_.startswith('ge.'))
    for k_, t in enc.items():
        assert_grad_close(t.grad, fx['ge.' + k_], tol=5e-4, floor=floor, name=k_)
    for k_, t in gen.items():
        if ('gd.' + k_) in fx:
            assert_grad_close(t.grad, fx['gd.' + k_], tol=5e-4, name=k_)


def test_epoch_two_steps():
    """train_epoch (train_mnist.py:300-346): running means and post-Adam parameters."""
    fx = load_golden('epoch_2steps')
    cfg, n = step_cfg(fx)
    enc = tdict(fx, 'e.', requires_grad=True)
    gen = tdict(fx, 'd.', requires_grad=True)
    st = O.new_opt_state(enc, gen)
    data = torch.from_numpy(fx['data'])
    c = 0
    acc = [0.0, 0.0, 0.0]
    for i in range(2):
        noise = dict(E=torch.from_numpy(fx[f'E{i}']), eps_z=torch.from_numpy(fx[f'eps_z{i}']),
                     eps_theta=torch.from_numpy(fx[f'eps_theta{i}']))
        e, lp, kl = O.train_step(O.image_coords(n), data[4 * i:4 * i + 4], enc, gen, st, noise,
                                 likelihood='bce', **cfg)
        c += 4
        for j, v in enumerate((e, -lp, kl)):
            acc[j] += 4 * (v - acc[j]) / c
    assert abs(acc[0] - float(fx['elbo'])) / abs(float(fx['elbo'])) < 1e-6
    assert abs(acc[1] - float(fx['err'])) / abs(float(fx['err'])) < 1e-6
    assert abs(acc[2] - float(fx['kl'])) / abs(float(fx['kl'])) < 1e-5
    for k_, t in enc.items():
        if k_ == 'conv_a.bias':   # analytic grad is 0 (softmax shift invariance): Adam follows rounding noise
            assert (t - torch.from_numpy(fx['e1.' + k_])).abs().max() <= 2 * 2e-4 * 2 + 1e-7
            continue
        assert rel_err(t, fx['e1.' + k_]) < 1e-5, k_
    for k_, t in gen.items():
        assert rel_err(t, fx['d1.' + k_]) < 1e-5, k_


def test_trajectory_20_steps():
    """20 consecutive reference Adam steps (train_mnist.py:300-346, lr 2e-3): the per-step ELBO / Error / KL curve and the
    final parameters.  The curve is held to 1e-5 (KL 1e-4).  The parameters are compared through the 20-step UPDATE
    p_T - p_0 (conftest.assert_trajectory_update_close): Adam normalises every element's step to ~lr whatever the size
    of its gradient, so elements whose gradient is rounding noise (dead hidden units, filter corners) walk +-lr per step
    in a direction no two fp32 evaluations agree on -- measured between the reference and this oracle, both on the CPU:
    relative L2 error of the update up to 9.4e-3 (decoder first layers), cosine >= 0.99995.  Gate: 2e-2 / cosine 0.999."""
    fx = load_golden('trajectory_20steps')
    cfg, n = step_cfg(fx)
    enc = tdict(fx, 'e.', requires_grad=True)
    gen = tdict(fx, 'd.', requires_grad=True)
    st = O.new_opt_state(enc, gen, lr=float(fx['lr']))
    data = torch.from_numpy(fx['data'])
    T, B = fx['curve'].shape[0], fx['E'].shape[1]
    for t in range(T):
        noise = dict(E=torch.from_numpy(fx['E'][t]), eps_z=torch.from_numpy(fx['eps_z'][t]),
                     eps_theta=torch.from_numpy(fx['eps_theta'][t]))
        e, lp, kl = O.train_step(O.image_coords(n), data[B * t:B * t + B], enc, gen, st, noise, likelihood='bce', **cfg)
        want = fx['curve'][t]
        assert abs(e - want[0]) / abs(want[0]) < 1e-5 and abs(-lp - want[1]) / abs(want[1]) < 1e-5, (t, e, lp, want)
        assert abs(kl - want[2]) / abs(want[2]) < 1e-4, (t, kl, want)
    assert_trajectory_update_close(enc, gen, fx, 2e-2)


def test_particles_tail_wide():
    """CTF + mask tail at the reference's default widths (128 kernels, hidden 512), seed-based fixture."""
    from conftest import seeded_models
    fx = load_golden('wide_particles32_ctf_mask')
    cfg, n = step_cfg(fx)
    enc_m, gen_m, _ = seeded_models(fx)
    enc = {k_: v.detach().clone().requires_grad_(True) for k_, v in enc_m.state_dict().items()}
    gen = {k_: v.detach().clone().requires_grad_(True) for k_, v in gen_m.state_dict().items()}
    elbo, logp, kl = O.elbo_step(O.image_coords(n), torch.from_numpy(fx['y']), enc, gen, likelihood='gauss',
                                 E=torch.from_numpy(fx['E']), eps_z=torch.from_numpy(fx['eps_z']),
                                 eps_theta=torch.from_numpy(fx['eps_theta']), ctf=torch.from_numpy(fx['ctf']),
                                 mask_radius=int(fx['mask_radius']), **cfg)
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < 1e-6
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < 1e-6
    (-elbo).backward()
    floor = 1e-3 * max(float(np.abs(v).max()) for k_, v in fx.items() if k_.startswith('ge.'))
    for k_, t in enc.items():
        assert_grad_close(t.grad, fx['ge.' + k_], tol=5e-4, floor=floor, name=k_)
    for k_, t in gen.items():
        assert_grad_close(t.grad, fx['gd.' + k_], tol=5e-4, name=k_)


PART_TAIL = ['step_particles32_ctf', 'step_particles32_mask', 'step_particles32_ctf_mask']


@pytest.mark.parametrize('name', PART_TAIL)
def test_particles_tail(name):
    """CTF filter + circular mask likelihood tail (train_particles.py:298-338)."""
    fx = load_golden(name)
    cfg, n = step_cfg(fx)
    enc = tdict(fx, 'e.', requires_grad=True)
    gen = tdict(fx, 'd.', requires_grad=True)
    ctf = torch.from_numpy(fx['ctf']) if 'ctf' in fx else None
    elbo, logp, kl = O.elbo_step(O.image_coords(n), torch.from_numpy(fx['y']), enc, gen, likelihood='gauss',
                                 E=torch.from_numpy(fx['E']), eps_z=torch.from_numpy(fx['eps_z']),
                                 eps_theta=torch.from_numpy(fx['eps_theta']), ctf=ctf,
                                 mask_radius=int(fx['mask_radius']), **cfg)
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < 1e-6
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < 1e-6
    (-elbo).backward()
    floor = 1e-3 * max(float(np.abs(v).max()) for k_, v in fx.items() if k_.startswith('ge.'))
    for k_, t in enc.items():
        assert_grad_close(t.grad, fx['ge.' + k_], floor=floor, name=k_)
    for k_, t in gen.items():
        if ('gd.' + k_) in fx:
            assert_grad_close(t.grad, fx['gd.' + k_], name=k_)


def test_ctf_filters():
    """Real-space CTF kernels (src/ctf.py:32-55)."""
    fx = load_golden('ctf_filters')
    n = int(fx['n'])
    got = O.ctf_filters(fx['defocus'], fx['cs'], fx['voltage'], fx['apix'], fx['bfactor'], fx['ampcont'],
                        fx['dfang'], n, n)
    assert rel_err(got, fx['filters']) < 1e-6


@pytest.mark.parametrize('name', ['get_latent_P8_28', 'get_latent_P4_20_norefine'])
def test_get_latent(name):
    fx = load_golden(name)
    n, cin, zd, C, k, p, R, refine, normal = [int(v) for v in fx['cfg']]
    zc, th, dx = O.get_latent(O.image_coords(n), torch.from_numpy(fx['y']), tdict(fx, 'p.'), R, p, bool(refine),
                              float(fx['theta_prior']), bool(normal))
    assert rel_err(zc, fx['z_content']) < TOL and rel_err(th, fx['theta_mu']) < TOL and rel_err(dx, fx['dx']) < TOL
