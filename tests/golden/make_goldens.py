#!/usr/bin/env python3
"""Generate the committed golden fixtures (tests/golden/*.npz) from the REAL reference.

Runs only in the build container, where the reference is mounted read-only at
/root/reference.  It imports the reference's `src.models`, `train_mnist`, `train_particles`
and `train_galaxy` (torchvision is stubbed: it is only used inside `main`), runs them on
the CPU with fixed seeds, and stores inputs, parameters, the injected noise and every
output / gradient as small float arrays.  No reference source is copied; fixtures are data.

Noise replay: the reference draws, per step, `empty(B,P).exponential_()` (models.py:387),
`normal(B,z,1)` (train_mnist.py:206) and `normal(B,1,1)` (train_mnist.py:230) in that order;
we pre-draw the same three tensors under the same seed and store them.

Usage:  python tests/golden/make_goldens.py        (writes next to this file)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import torch

torch.set_num_threads(8)


def _import_reference():
    sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
    sys.path.insert(0, REF)
    import src.models as models          # noqa
    import train_mnist                   # noqa
    import train_particles               # noqa
    import train_galaxy                  # noqa
    sys.path.pop(0)
    return models, train_mnist, train_particles, train_galaxy


models, train_mnist, train_particles, train_galaxy = _import_reference()


def npd(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **npd(arrs))
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def coords(n):
    xg = np.linspace(-1, 1, n)
    yg = np.linspace(1, -1, n)
    x0, x1 = np.meshgrid(xg, yg)
    return torch.from_numpy(np.stack([x0.ravel(), x1.ravel()], 1)).float()


def draw_noise(seed, B, P, zd):
    torch.manual_seed(seed)
    E = torch.empty(B, P).exponential_()
    eps_z = torch.normal(torch.zeros(B, zd, 1), torch.ones(B, zd, 1))
    eps_t = torch.normal(torch.zeros(B, 1, 1), torch.ones(B, 1, 1))
    return E, eps_z.view(B, zd), eps_t.view(B)


# ---------------------------------------------------------------------------------------
def gen_bank():
    for k in (5, 28):
        for R in (4, 8, 16):
            for Cin in (1, 3):
                torch.manual_seed(k * 100 + R * 10 + Cin)
                gc = models.GroupConv(Cin, 3, k, padding=0, input_rot_dim=1, output_rot_dim=R)
                tw = gc.trans_filter('cpu')
                gup = torch.randn_like(tw)
                (tw * gup).sum().backward()
                save(f'bank_k{k}_R{R}_Cin{Cin}', weight=gc.weight, bank=tw, gbank=gup,
                     gweight=gc.weight.grad)


def gen_groupconv():
    torch.manual_seed(7)
    gc = models.GroupConv(1, 8, 28, padding=8, input_rot_dim=1, output_rot_dim=8)
    y = torch.rand(2, 1, 28, 28)
    out = gc(y, 'cpu')
    gout = torch.randn_like(out)
    (out * gout).sum().backward()
    save('groupconv_fwd_bwd', y=y, weight=gc.weight, bias=gc.bias, out=out, gout=gout,
         gweight=gc.weight.grad, gbias=gc.bias.grad)
    # 3 input channels, odd kernel, R=4
    torch.manual_seed(8)
    gc = models.GroupConv(3, 4, 9, padding=3, input_rot_dim=1, output_rot_dim=4)
    y = torch.rand(3, 3, 12, 12)
    out = gc(y, 'cpu')
    gout = torch.randn_like(out)
    (out * gout).sum().backward()
    save('groupconv_cin3_k9_R4', y=y, weight=gc.weight, bias=gc.bias, out=out, gout=gout,
         gweight=gc.weight.grad, gbias=gc.bias.grad)


def _enc(n, cin, zd, C, k, p, R, refine, theta_prior, normal, seed, scale_heads=1.0):
    torch.manual_seed(seed)
    enc = models.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, cin, zd, kernels_num=C, kernels_size=k, padding=p, groupconv=R,
        rot_refinement=refine, theta_prior=theta_prior, normal_prior_over_r=normal)
    if scale_heads != 1.0:
        with torch.no_grad():
            for nm in ('conv_a', 'conv_r', 'conv_z'):
                getattr(enc, nm).weight.mul_(scale_heads)
    return enc


def gen_encoder():
    cases = [
        ('encoder_P8_28', dict(n=28, cin=1, zd=2, C=128, k=28, p=8, R=8, refine=True,
                               theta_prior=np.pi, normal=False, seed=0), 2, 'rand'),
        ('encoder_P16_28_normal', dict(n=28, cin=1, zd=2, C=16, k=28, p=8, R=16, refine=True,
                                       theta_prior=np.pi / 4, normal=True, seed=1,
                                       scale_heads=20.0), 3, 'rand'),
        ('encoder_P4_20_norefine', dict(n=20, cin=1, zd=3, C=16, k=20, p=4, R=4, refine=False,
                                        theta_prior=np.pi, normal=False, seed=2,
                                        scale_heads=20.0), 2, 'randn'),
        ('encoder_P8_64', dict(n=64, cin=1, zd=2, C=16, k=64, p=16, R=8, refine=True,
                               theta_prior=np.pi, normal=False, seed=3, scale_heads=10.0), 2, 'randn'),
    ]
    for name, kw, B, data in cases:
        enc = _enc(**kw)
        torch.manual_seed(100)
        y = torch.rand(B, kw['cin'], kw['n'], kw['n']) if data == 'rand' else \
            torch.randn(B, kw['cin'], kw['n'], kw['n'])
        Ho = kw['n'] + 2 * kw['p'] - kw['k'] + 1
        P = kw['R'] * Ho * Ho
        E, _, _ = draw_noise(123, B, P, kw['zd'])
        torch.manual_seed(123)
        attn, q, p_r, a_s, offs, theta, z = enc(y, 'cpu')
        # scalar probe for gradients through every differentiable output
        torch.manual_seed(5)
        w_q, w_a, w_t, w_z = (torch.randn_like(q), torch.randn_like(a_s), torch.randn_like(theta),
                              torch.randn_like(z))
        probe = (q * w_q).sum() + (a_s * w_a).sum() * 50 + (theta * w_t).sum() + (z * w_z).sum() \
            + (attn * w_q).sum() * 0.5
        probe.backward()
        out = dict(y=y, E=E, attn=attn, q_t_r=q, p_r=p_r, a_sampled=a_s, offsets=offs, theta=theta,
                   z=z, w_q=w_q, w_a=w_a, w_t=w_t, w_z=w_z,
                   cfg=np.array([kw['n'], kw['cin'], kw['zd'], kw['C'], kw['k'], kw['p'], kw['R'],
                                 int(kw['refine']), int(kw['normal'])]),
                   theta_prior=np.float64(kw['theta_prior']))
        for k_, v in enc.state_dict().items():
            out['p.' + k_] = v
        for k_, v in enc.named_parameters():
            out['g.' + k_] = v.grad
        save(name, **out)


def gen_decoder():
    cases = [
        ('decoder_plain', dict(latent_dim=2, hidden_dim=64, n_out=1, num_layers=2), 2, 64),
        ('decoder_plain512', dict(latent_dim=2, hidden_dim=512, n_out=1, num_layers=2), 2, 49),
        ('decoder_fourier', dict(latent_dim=2, hidden_dim=64, n_out=1, num_layers=2,
                                 fourier_expansion=True, sigma=2.0 / 27), 2, 100),
        ('decoder_resid', dict(latent_dim=3, hidden_dim=64, n_out=1, num_layers=3, resid=True), 2, 64),
        ('decoder_nout2', dict(latent_dim=2, hidden_dim=64, n_out=2, num_layers=2), 2, 64),
        ('decoder_nout3_z50_L4', dict(latent_dim=50, hidden_dim=64, n_out=3, num_layers=4,
                                      fourier_expansion=True, sigma=2.0 / 31), 2, 81),
        ('decoder_z0_L1', dict(latent_dim=0, hidden_dim=64, n_out=1, num_layers=1), 2, 36),
    ]
    for name, kw, B, N in cases:
        torch.manual_seed(11)
        gen = models.SpatialGenerator(**kw)
        torch.manual_seed(12)
        x = (torch.rand(B, N, 2) * 2 - 1).requires_grad_(True)
        z = torch.randn(B, max(kw['latent_dim'], 1)).requires_grad_(True) if kw['latent_dim'] > 0 else None
        yh = gen(x, z)
        gy = torch.randn_like(yh)
        (yh * gy).sum().backward()
        out = dict(x=x, y_hat=yh, gy=gy, gx=x.grad,
                   cfg=np.array([kw['latent_dim'], kw['hidden_dim'], kw['n_out'], kw['num_layers'],
                                 int(kw.get('resid', False)), int(kw.get('fourier_expansion', False))]),
                   sigma=np.float64(kw.get('sigma', 0.0)))
        if z is not None:
            out['z'] = z
            out['gz'] = z.grad
        for k_, v in gen.state_dict().items():
            out['p.' + k_] = v
        for k_, v in gen.named_parameters():
            out['g.' + k_] = v.grad
        save(name, **out)


def _step_case(name, tm, *, n, cin, zd, C, k, p, R, r_inf, dataset_normal, hidden, layers, n_out,
               fourier, resid, B, data, scale_heads, particles=False, seed=0):
    torch.manual_seed(seed)
    sigma = 2.0 / (n - 1)
    gen = models.SpatialGenerator(zd, hidden, n_out=n_out, num_layers=layers, resid=resid,
                                  fourier_expansion=fourier, sigma=sigma)
    theta_prior = np.pi / 4 if dataset_normal else np.pi
    enc = models.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, cin, zd, kernels_num=C, kernels_size=k, padding=p, groupconv=R,
        rot_refinement=(r_inf == 'attention+offsets'), theta_prior=theta_prior,
        normal_prior_over_r=dataset_normal)
    if scale_heads != 1.0:
        with torch.no_grad():
            for nm in ('conv_a', 'conv_r', 'conv_z'):
                getattr(enc, nm).weight.mul_(scale_heads)
    torch.manual_seed(0)
    y = torch.rand(B, cin, n, n) if data == 'rand' else torch.randn(B, cin, n, n)
    x_coord = coords(n)
    Ho = n + 2 * p - k + 1
    P = R * Ho * Ho
    E, eps_z, eps_t = draw_noise(123, B, P, zd)
    torch.manual_seed(123)
    if particles:
        elbo, logp, kl = tm.eval_minibatch(x_coord, y, None, gen, enc, 'attention', r_inf, 0, 'cpu',
                                           theta_prior, R, p, 0)
    else:
        elbo, logp, kl = tm.eval_minibatch(x_coord, y, gen, enc, 'attention', r_inf, 0, 'cpu',
                                           theta_prior, R, n)
    (-elbo).backward()
    out = dict(y=y, E=E, eps_z=eps_z, eps_theta=eps_t, elbo=elbo, log_p=logp, kl=kl,
               cfg=np.array([n, cin, zd, C, k, p, R, int(r_inf == 'attention+offsets'),
                             int(dataset_normal), hidden, layers, n_out, int(fourier), int(resid)]),
               theta_prior=np.float64(theta_prior), sigma=np.float64(sigma))
    for k_, v in enc.state_dict().items():
        out['e.' + k_] = v
    for k_, v in gen.state_dict().items():
        out['d.' + k_] = v
    for k_, v in enc.named_parameters():
        out['ge.' + k_] = v.grad
    for k_, v in gen.named_parameters():
        out['gd.' + k_] = v.grad
    save(name, **out)
    print('   ', name, 'elbo', float(elbo), 'log_p', float(logp), 'kl', float(kl), elbo.dtype, logp.dtype)


def gen_steps():
    common = dict(hidden=512, layers=2, n_out=1, fourier=False, resid=False)
    _step_case('step_mnist28_P8_init', train_mnist, n=28, cin=1, zd=2, C=128, k=28, p=8, R=8,
               r_inf='attention+offsets', dataset_normal=False, B=4, data='rand', scale_heads=1.0, **common)
    _step_case('step_mnist28_P8_peaked', train_mnist, n=28, cin=1, zd=2, C=32, k=28, p=8, R=8,
               r_inf='attention+offsets', dataset_normal=False, B=4, data='rand', scale_heads=25.0,
               hidden=128, layers=2, n_out=1, fourier=False, resid=False)
    _step_case('step_mnist28_P16_fourier_normal', train_mnist, n=28, cin=1, zd=2, C=16, k=28, p=8, R=16,
               r_inf='attention+offsets', dataset_normal=True, B=3, data='rand', scale_heads=25.0,
               hidden=64, layers=2, n_out=1, fourier=True, resid=False)
    _step_case('step_mnist28_P4_attention_resid', train_mnist, n=28, cin=1, zd=3, C=16, k=28, p=8, R=4,
               r_inf='attention', dataset_normal=False, B=3, data='rand', scale_heads=25.0,
               hidden=64, layers=3, n_out=1, fourier=False, resid=True)
    _step_case('step_particles64_P8', train_particles, n=64, cin=1, zd=2, C=32, k=64, p=16, R=8,
               r_inf='attention+offsets', dataset_normal=False, B=2, data='randn', scale_heads=10.0,
               hidden=128, layers=2, n_out=1, fourier=False, resid=False, particles=True)
    _step_case('step_particles32_fitnoise', train_particles, n=32, cin=1, zd=2, C=16, k=32, p=8, R=8,
               r_inf='attention+offsets', dataset_normal=False, B=2, data='randn', scale_heads=10.0,
               hidden=64, layers=2, n_out=2, fourier=False, resid=False, particles=True)
    _step_case('step_galaxy_small', train_galaxy, n=32, cin=3, zd=50, C=16, k=16, p=8, R=16,
               r_inf='attention+offsets', dataset_normal=False, B=2, data='rand', scale_heads=10.0,
               hidden=64, layers=4, n_out=3, fourier=True, resid=False)


def gen_particles_tail():
    """CTF filter generation (src/ctf.py) and the CTF / mask likelihood tail (train_particles.py:298-338)."""
    import pandas as pd
    sys.path.insert(0, REF)
    import src.ctf as C
    sys.path.pop(0)
    rng = np.random.RandomState(0)
    nrow = 3
    params = pd.DataFrame(dict(defocus=rng.uniform(1.0, 3.0, nrow), cs=np.full(nrow, 2.7),
                               voltage=np.full(nrow, 300.0), apix=np.full(nrow, 1.2),
                               bfactor=rng.uniform(50, 150, nrow), ampcont=np.full(nrow, 7.0),
                               dfdiff=np.zeros(nrow), dfang=rng.uniform(0, 180, nrow)))
    n = 32
    filt = C.ctf_filter(params, n - 1, n - 1, scale=1)
    save('ctf_filters', filters=filt, **{k_: params[k_].values for k_ in params.columns}, n=np.int64(n - 1))
    for name, use_ctf, radius in (('step_particles32_ctf', True, 0), ('step_particles32_mask', False, 9),
                                  ('step_particles32_ctf_mask', True, 11)):
        torch.manual_seed(4)
        zd, C_, k, p, R, B = 2, 16, 32, 8, 8, 3
        gen = models.SpatialGenerator(zd, 64, n_out=1, num_layers=2)
        enc = models.InferenceNetwork_AttentionTranslation_AttentionRotation(
            n, 1, zd, kernels_num=C_, kernels_size=k, padding=p, groupconv=R, rot_refinement=True,
            theta_prior=np.pi, normal_prior_over_r=False)
        with torch.no_grad():
            for nm in ('conv_a', 'conv_r', 'conv_z'):
                getattr(enc, nm).weight.mul_(10.0)
        torch.manual_seed(0)
        y = torch.randn(B, 1, n, n)
        ctf = torch.from_numpy(filt[:B]).float().unsqueeze(1) if use_ctf else None
        Ho = n + 2 * p - k + 1
        E, eps_z, eps_t = draw_noise(123, B, R * Ho * Ho, zd)
        torch.manual_seed(123)
        elbo, logp, kl = train_particles.eval_minibatch(coords(n), y, ctf, gen, enc, 'attention',
                                                        'attention+offsets', 0, 'cpu', np.pi, R, p, radius)
        (-elbo).backward()
        out = dict(y=y, E=E, eps_z=eps_z, eps_theta=eps_t, elbo=elbo, log_p=logp, kl=kl,
                   cfg=np.array([n, 1, zd, C_, k, p, R, 1, 0, 64, 2, 1, 0, 0]), theta_prior=np.float64(np.pi),
                   sigma=np.float64(2.0 / (n - 1)), mask_radius=np.int64(radius))
        if use_ctf:
            out['ctf'] = ctf
        for k_, v in enc.state_dict().items():
            out['e.' + k_] = v
        for k_, v in gen.state_dict().items():
            out['d.' + k_] = v
        for k_, v in enc.named_parameters():
            out['ge.' + k_] = v.grad
        for k_, v in gen.named_parameters():
            out['gd.' + k_] = v.grad
        save(name, **out)
        print('   ', name, float(elbo), float(logp), float(kl))


def gen_get_latent():
    """clustering_mnist.get_latent (the REAL function; plotting / astropy imports are stubbed)."""
    class _Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            m = _Stub(self.__name__ + '.' + k)
            setattr(self, k, m)
            return m

        def __call__(self, *a, **k):
            return None
    for nm in ('seaborn', 'astropy', 'astropy.stats', 'astropy.units', 'matplotlib', 'matplotlib.pyplot',
               'matplotlib.cm', 'matplotlib.colors', 'mpl_toolkits', 'mpl_toolkits.mplot3d'):
        try:
            __import__(nm)
        except Exception:
            sys.modules[nm] = _Stub(nm)
    sys.path.insert(0, REF)
    import clustering_mnist as cm
    sys.path.pop(0)
    for name, kw, B, scale in (('get_latent_P8_28', dict(n=28, cin=1, zd=2, C=16, k=28, p=8, R=8, refine=True,
                                                          theta_prior=np.pi, normal=False, seed=5), 4, 20.0),
                               ('get_latent_P4_20_norefine', dict(n=20, cin=1, zd=3, C=16, k=20, p=4, R=4, refine=False,
                                                                   theta_prior=np.pi, normal=False, seed=6), 3, 20.0)):
        enc = _enc(scale_heads=scale, **kw)
        torch.manual_seed(9)
        y = torch.rand(B, kw['cin'], kw['n'], kw['n'])
        r_inf = 'attention+offsets' if kw['refine'] else 'attention'
        zc, th, dx = cm.get_latent(coords(kw['n']), y, enc, 'attention', r_inf, 'cpu', kw['n'])
        out = dict(y=y, z_content=zc, theta_mu=th, dx=dx,
                   cfg=np.array([kw['n'], kw['cin'], kw['zd'], kw['C'], kw['k'], kw['p'], kw['R'], int(kw['refine']),
                                 int(kw['normal'])]), theta_prior=np.float64(kw['theta_prior']))
        for k_, v in enc.state_dict().items():
            out['p.' + k_] = v
        save(name, **out)


def gen_mrc():
    """A small stack written by the reference's src/mrc.write (fixture = data file), plus one with an extended header."""
    sys.path.insert(0, REF)
    import src.mrc as mrc
    sys.path.pop(0)
    rng = np.random.RandomState(3)
    arr = rng.randn(5, 6, 7).astype(np.float32)
    with open(os.path.join(HERE, 'stack_ref.mrcs'), 'wb') as f:
        mrc.write(f, arr, ax=1.5, ay=2.5, az=3.5)
    with open(os.path.join(HERE, 'stack_ref_ext.mrc'), 'wb') as f:
        hdr = mrc.make_header((1, 6, 7), (1, 1, 1), (90, 90, 90), dtype=np.int16, exthd_size=16)
        mrc.write(f, (arr[0] * 100).astype(np.int16), header=hdr, extended_header=b'0123456789abcdef')
    save('mrc_arrays', stack=arr, single=(arr[0] * 100).astype(np.int16))
    print('wrote mrc fixtures')


def gen_secondary():
    """The two secondary branches of eval_minibatch (train_mnist.py:35-185): unimodal/unimodal (MLP encoder) and
    attention/unimodal (translation attention, rotation pooled by fc_r or plain conv)."""
    n, zd, B = 20, 2, 3
    x_coord = coords(n)
    torch.manual_seed(0)
    y = torch.rand(B, 1, n, n)
    cases = [('step_unimodal_unimodal', 'unimodal', 'unimodal', 0), ('step_attention_unimodal_gc4', 'attention', 'unimodal', 4),
             ('step_attention_unimodal_gc0', 'attention', 'unimodal', 0)]
    for name, t_inf, r_inf, gc in cases:
        torch.manual_seed(1)
        gen = models.SpatialGenerator(zd, 32, num_layers=2)
        if t_inf == 'unimodal':
            enc = models.InferenceNetwork_UnimodalTranslation_UnimodalRotation(n * n, zd + 3, 32, num_layers=2)
        else:
            enc = models.InferenceNetwork_AttentionTranslation_UnimodalRotation(n, 1, zd, kernels_num=8, groupconv=gc)
            with torch.no_grad():
                for nm in ('conv_a', 'conv_r', 'conv_z'):
                    getattr(enc, nm).weight.mul_(10.0)
        torch.manual_seed(77)
        state = torch.get_rng_state()
        elbo, logp, kl = train_mnist.eval_minibatch(x_coord, y, gen, enc, t_inf, r_inf, 0, 'cpu', np.pi, gc, n)
        (-elbo).backward()
        out = dict(y=y, elbo=elbo, log_p=logp, kl=kl, rng_seed=np.int64(77), cfg=np.array([n, zd, gc]))
        for k_, v in enc.state_dict().items():
            out['e.' + k_] = v
        for k_, v in gen.state_dict().items():
            out['d.' + k_] = v
        for k_, v in enc.named_parameters():
            out['ge.' + k_] = v.grad
        for k_, v in gen.named_parameters():
            out['gd.' + k_] = v.grad
        # replay of the reference's random draws under the same seed, in its order
        torch.manual_seed(77)
        if t_inf == 'unimodal':
            out['eps'] = y.new(B, zd + 3).normal_()
        else:
            Ho = n + 2 * (n // 2) - n + 1
            out['E'] = torch.empty(B, Ho * Ho).exponential_()
            out['eps_z'] = torch.normal(torch.zeros(B, zd, 1), torch.ones(B, zd, 1)).view(B, zd)
            out['eps_theta'] = torch.normal(torch.zeros(B, 1, 1), torch.ones(B, 1, 1)).view(B)
        save(name, **out)
        print('   ', name, float(elbo), float(logp), float(kl), elbo.dtype)


def gen_epoch():
    """train_epoch over 2 minibatches (train_mnist.py:300-346): running means + post-Adam params."""
    torch.manual_seed(0)
    n, zd, C, R, p, k = 28, 2, 16, 8, 8, 28
    gen = models.SpatialGenerator(zd, 64, num_layers=2)
    enc = models.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, 1, zd, kernels_num=C, kernels_size=k, padding=p, groupconv=R, rot_refinement=True,
        theta_prior=np.pi, normal_prior_over_r=False)
    torch.manual_seed(0)
    data = torch.rand(8, 1, n, n)
    x_coord = coords(n)
    params = list(gen.parameters()) + list(enc.parameters())
    optim = torch.optim.Adam(params, lr=2e-4)
    Ho = n + 2 * p - k + 1
    P = R * Ho * Ho
    init = {('e.' + k_): v.clone() for k_, v in enc.state_dict().items()}
    init.update({('d.' + k_): v.clone() for k_, v in gen.state_dict().items()})
    # noise for the two steps, drawn back-to-back from one seeded stream like the reference does
    torch.manual_seed(321)
    noises = []
    for _ in range(2):
        E = torch.empty(4, P).exponential_()
        ez = torch.normal(torch.zeros(4, zd, 1), torch.ones(4, zd, 1)).view(4, zd)
        et = torch.normal(torch.zeros(4, 1, 1), torch.ones(4, 1, 1)).view(4)
        noises.append((E, ez, et))
    it = [(data[0:4],), (data[4:8],)]
    torch.manual_seed(321)
    elbo, err, kl = train_mnist.train_epoch(it, x_coord, gen, enc, optim, 'attention', 'attention+offsets',
                                            0, 1, 8, 'cpu', params, np.pi, R, n)
    out = dict(data=data, elbo=np.float64(elbo), err=np.float64(err), kl=np.float64(kl),
               cfg=np.array([n, 1, zd, C, k, p, R, 1, 0, 64, 2, 1, 0, 0]),
               theta_prior=np.float64(np.pi), sigma=np.float64(2.0 / (n - 1)))
    for i, (E, ez, et) in enumerate(noises):
        out[f'E{i}'], out[f'eps_z{i}'], out[f'eps_theta{i}'] = E, ez, et
    out.update(init)
    for k_, v in enc.state_dict().items():
        out['e1.' + k_] = v
    for k_, v in gen.state_dict().items():
        out['d1.' + k_] = v
    save('epoch_2steps', **out)
    print('    epoch', elbo, err, kl)


def gen_trajectory():
    """20 consecutive reference Adam steps (train_mnist.py:300-346 called once per minibatch so that the per-step ELBO is
    observable): 28x28, P8, C = 16, hidden 64, 4 images per step, lr 2e-3 so that the parameters really move.  Stores the
    data, the initial state, the noise of every step, the ELBO / Error / KL curve and the final parameters."""
    torch.manual_seed(0)
    n, zd, C, R, p, k, B, T = 28, 2, 16, 8, 8, 28, 4, 20
    gen = models.SpatialGenerator(zd, 64, num_layers=2)
    enc = models.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, 1, zd, kernels_num=C, kernels_size=k, padding=p, groupconv=R, rot_refinement=True,
        theta_prior=np.pi, normal_prior_over_r=False)
    with torch.no_grad():
        for nm in ('conv_a', 'conv_r', 'conv_z'):
            getattr(enc, nm).weight.mul_(10.0)
    torch.manual_seed(1)
    data = torch.rand(B * T, 1, n, n)
    x_coord = coords(n)
    params = list(gen.parameters()) + list(enc.parameters())
    lr = 2e-3
    optim = torch.optim.Adam(params, lr=lr)
    Ho = n + 2 * p - k + 1
    P = R * Ho * Ho
    out = dict(data=data, lr=np.float64(lr), cfg=np.array([n, 1, zd, C, k, p, R, 1, 0, 64, 2, 1, 0, 0]),
               theta_prior=np.float64(np.pi), sigma=np.float64(2.0 / (n - 1)))
    out.update({('e.' + k_): v.clone() for k_, v in enc.state_dict().items()})
    out.update({('d.' + k_): v.clone() for k_, v in gen.state_dict().items()})
    torch.manual_seed(555)                       # pre-draw the noise of all steps from one stream, in the reference's order
    Es, ezs, ets = [], [], []
    for _ in range(T):
        Es.append(torch.empty(B, P).exponential_())
        ezs.append(torch.normal(torch.zeros(B, zd, 1), torch.ones(B, zd, 1)).view(B, zd))
        ets.append(torch.normal(torch.zeros(B, 1, 1), torch.ones(B, 1, 1)).view(B))
    torch.manual_seed(555)
    curve = []
    for t in range(T):
        e, err, kl = train_mnist.train_epoch([(data[t * B:(t + 1) * B],)], x_coord, gen, enc, optim, 'attention',
                                             'attention+offsets', 0, 1, B, 'cpu', params, np.pi, R, n)
        curve.append((e, err, kl))
    out['E'], out['eps_z'], out['eps_theta'] = torch.stack(Es), torch.stack(ezs), torch.stack(ets)
    out['curve'] = np.array(curve, dtype=np.float64)
    for k_, v in enc.state_dict().items():
        out['eT.' + k_] = v
    for k_, v in gen.state_dict().items():
        out['dT.' + k_] = v
    save('trajectory_20steps', **out)
    print('    trajectory ELBO', [round(c_[0], 2) for c_ in curve])


def gen_wide():
    """Secondary encoder (translation attention, rotation pooled by fc_r; src/models.py:268-319) and the particle CTF + mask
    tail (train_particles.py:298-338) at the reference's DEFAULT widths (128 kernels, hidden 512), seed-based like the
    hot-path fixtures: parameters are re-created from the seed by the drop-in classes and checked against digests."""
    # (a) attention / unimodal, groupconv 4, C = 128, hidden 512 (train_mnist.py:86-185)
    n, zd, B, gc, seed = 20, 2, 3, 4, 51
    torch.manual_seed(seed)
    gen = models.SpatialGenerator(zd, 512, num_layers=2)
    enc = models.InferenceNetwork_AttentionTranslation_UnimodalRotation(n, 1, zd, kernels_num=128, groupconv=gc)
    with torch.no_grad():
        for nm in ('conv_a', 'conv_r', 'conv_z'):
            getattr(enc, nm).weight.mul_(10.0)
    torch.manual_seed(seed + 1000)
    y = torch.rand(B, 1, n, n)
    torch.manual_seed(77)
    elbo, logp, kl = train_mnist.eval_minibatch(coords(n), y, gen, enc, 'attention', 'unimodal', 0, 'cpu', np.pi, gc, n)
    (-elbo).backward()
    out = dict(y=y, elbo=elbo, log_p=logp, kl=kl, cfg=np.array([n, zd, gc, 128, 512]), seed=np.int64(seed),
               scale_heads=np.float64(10.0))
    torch.manual_seed(77)
    Ho = n + 2 * (n // 2) - n + 1
    out['E'] = torch.empty(B, Ho * Ho).exponential_()
    out['eps_z'] = torch.normal(torch.zeros(B, zd, 1), torch.ones(B, zd, 1)).view(B, zd)
    out['eps_theta'] = torch.normal(torch.zeros(B, 1, 1), torch.ones(B, 1, 1)).view(B)
    for pre, mod in (('e.', enc), ('d.', gen)):
        for k_, v in mod.state_dict().items():
            out['s' + pre + k_] = param_digest(v)
        for k_, v in mod.named_parameters():
            out['g' + pre + k_] = v.grad.clone()
    save('wide_attention_unimodal_gc4', **out)
    print('    wide_attention_unimodal_gc4', float(elbo), float(logp), float(kl))

    # (b) particle tail with CTF + circular mask, C = 128, hidden 512, 32x32
    import pandas as pd
    sys.path.insert(0, REF)
    import src.ctf as Cm
    sys.path.pop(0)
    rng = np.random.RandomState(1)
    nrow, n = 3, 32
    prm = pd.DataFrame(dict(defocus=rng.uniform(1.0, 3.0, nrow), cs=np.full(nrow, 2.7), voltage=np.full(nrow, 300.0),
                            apix=np.full(nrow, 1.2), bfactor=rng.uniform(50, 150, nrow), ampcont=np.full(nrow, 7.0),
                            dfdiff=np.zeros(nrow), dfang=rng.uniform(0, 180, nrow)))
    filt = Cm.ctf_filter(prm, n - 1, n - 1, scale=1)
    zd, k, p, R, B, seed, radius = 2, 32, 8, 8, 3, 52, 11
    torch.manual_seed(seed)
    gen = models.SpatialGenerator(zd, 512, n_out=1, num_layers=2)
    enc = models.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, 1, zd, kernels_num=128, kernels_size=k, padding=p, groupconv=R, rot_refinement=True, theta_prior=np.pi,
        normal_prior_over_r=False)
    with torch.no_grad():
        for nm in ('conv_a', 'conv_r', 'conv_z'):
            getattr(enc, nm).weight.mul_(10.0)
    torch.manual_seed(seed + 1000)
    y = torch.randn(B, 1, n, n)
    ctf = torch.from_numpy(filt[:B]).float().unsqueeze(1)
    Ho = n + 2 * p - k + 1
    E, eps_z, eps_t = draw_noise(123, B, R * Ho * Ho, zd)
    torch.manual_seed(123)
    elbo, logp, kl = train_particles.eval_minibatch(coords(n), y, ctf, gen, enc, 'attention', 'attention+offsets', 0,
                                                    'cpu', np.pi, R, p, radius)
    (-elbo).backward()
    out = dict(y=y, ctf=ctf, E=E, eps_z=eps_z, eps_theta=eps_t, elbo=elbo, log_p=logp, kl=kl,
               cfg=np.array([n, 1, zd, 128, k, p, R, 1, 0, 512, 2, 1, 0, 0]), theta_prior=np.float64(np.pi),
               sigma=np.float64(2.0 / (n - 1)), mask_radius=np.int64(radius), seed=np.int64(seed),
               scale_heads=np.float64(10.0))
    for pre, mod in (('e.', enc), ('d.', gen)):
        for k_, v in mod.state_dict().items():
            out['s' + pre + k_] = param_digest(v)
        for k_, v in mod.named_parameters():
            out['g' + pre + k_] = v.grad.clone()
    save('wide_particles32_ctf_mask', **out)
    print('    wide_particles32_ctf_mask', float(elbo), float(logp), float(kl))


def param_digest(t):
    """(sum, abs-sum, first, last) in float64: enough to detect any difference in seeded construction."""
    t = t.detach().double().reshape(-1)
    return np.array([float(t.sum()), float(t.abs().sum()), float(t[0]), float(t[-1])])


def gen_hotpath():
    """Full-WIDTH steps that reach the fused split-pipe decoder branches bench.py times (ops._dense_x6_ok + virt_act +
    virt + fuse_in; the Fourier [Wc | Wl] first layer): S64 widths (n=64 -> 4096 pixels per image, hidden 512, C=128,
    B=2) and S28F (n=28, P16, Fourier 1024 -> 512 -> 512 -> 1, B=8: 6272 pixels = 49 x 128).  The parameters are the
    reference's default init under a seed (generator first, then encoder, head weights scaled so that attention is
    peaked); the fixture stores the SEED and a digest of every state_dict entry instead of 3-4 MB of parameter arrays,
    plus the inputs, the injected noise, the three ELBO terms and EVERY parameter gradient from the real
    train_particles / train_mnist eval_minibatch."""
    cases = [
        ('hot_S64_B2', train_particles, dict(n=64, cin=1, zd=2, C=128, k=64, p=16, R=8, hidden=512, layers=2, n_out=1,
                                             fourier=False), 2, 'randn', 10.0, True, 31),
        ('hot_S28F_B8', train_mnist, dict(n=28, cin=1, zd=2, C=128, k=28, p=8, R=16, hidden=512, layers=2, n_out=1,
                                          fourier=True), 8, 'rand', 10.0, False, 32),
        # the reference's REAL MNIST-U / MNIST-N geometry: 50x50 images with the default 28-tap kernel and padding 8
        # (train_mnist.py:413-417) -> 39x39 outputs per rotation on a 66-wide frame
        ('hot_M50_B2', train_mnist, dict(n=50, cin=1, zd=2, C=128, k=28, p=8, R=8, hidden=512, layers=2, n_out=1,
                                         fourier=False), 2, 'rand', 10.0, False, 33),
    ]
    only = [a for a in sys.argv[2:] if a.startswith('hot_')]
    if only:
        cases = [c_ for c_ in cases if c_[0] in only]
    for name, tm, c, B, data, scale, particles, seed in cases:
        n, R, zd, p = c['n'], c['R'], c['zd'], c['p']
        sigma = 2.0 / (n - 1)
        torch.manual_seed(seed)
        gen = models.SpatialGenerator(zd, c['hidden'], n_out=c['n_out'], num_layers=c['layers'], resid=False,
                                      fourier_expansion=c['fourier'], sigma=sigma)
        enc = models.InferenceNetwork_AttentionTranslation_AttentionRotation(
            n, c['cin'], zd, kernels_num=c['C'], kernels_size=c['k'], padding=p, groupconv=R, rot_refinement=True,
            theta_prior=np.pi, normal_prior_over_r=False)
        with torch.no_grad():
            for nm in ('conv_a', 'conv_r', 'conv_z'):
                getattr(enc, nm).weight.mul_(scale)
        torch.manual_seed(seed + 1000)
        y = torch.rand(B, c['cin'], n, n) if data == 'rand' else torch.randn(B, c['cin'], n, n)
        Ho = n + 2 * p - c['k'] + 1
        E, eps_z, eps_t = draw_noise(123, B, R * Ho * Ho, zd)
        torch.manual_seed(123)
        if particles:
            elbo, logp, kl = tm.eval_minibatch(coords(n), y, None, gen, enc, 'attention', 'attention+offsets', 0,
                                               'cpu', np.pi, R, p, 0)
        else:
            elbo, logp, kl = tm.eval_minibatch(coords(n), y, gen, enc, 'attention', 'attention+offsets', 0, 'cpu',
                                               np.pi, R, n)
        (-elbo).backward()
        out = dict(y=y, E=E, eps_z=eps_z, eps_theta=eps_t, elbo=elbo, log_p=logp, kl=kl,
                   cfg=np.array([n, c['cin'], zd, c['C'], c['k'], p, R, 1, 0, c['hidden'], c['layers'], c['n_out'],
                                 int(c['fourier']), 0]),
                   theta_prior=np.float64(np.pi), sigma=np.float64(sigma), seed=np.int64(seed),
                   scale_heads=np.float64(scale))
        for k_, v in enc.state_dict().items():
            out['se.' + k_] = param_digest(v)
        for k_, v in gen.state_dict().items():
            out['sd.' + k_] = param_digest(v)
        for k_, v in enc.named_parameters():
            out['ge.' + k_] = v.grad.clone()
        for k_, v in gen.named_parameters():
            out['gd.' + k_] = v.grad.clone()
        # Conditioning of the gradients at the LeakyReLU kinks: with 3-6 M hidden activations per step a forward
        # difference of a few ulp moves a handful of pre-activations across 0, and each flip changes the derivative of
        # that element by 100x (measured: a 1e-7 relative change of the Fourier arguments flips 2 elements and moves
        # d/dtheta by 1.6e-3 of max-norm).  The fixture therefore carries the REFERENCE's own gradient change under a
        # 1e-5 relative input perturbation (two draws, max): `ke.*` / `kd.*`, in units of max|g|.
        sens = {}
        for trial in (1, 2):
            for q in list(enc.parameters()) + list(gen.parameters()):
                q.grad = None
            gper = torch.Generator().manual_seed(900 + trial)
            y2 = y * (1.0 + 1e-5 * torch.randn(y.shape, generator=gper))
            torch.manual_seed(123)
            if particles:
                e2, _, _ = tm.eval_minibatch(coords(n), y2, None, gen, enc, 'attention', 'attention+offsets', 0, 'cpu',
                                             np.pi, R, p, 0)
            else:
                e2, _, _ = tm.eval_minibatch(coords(n), y2, gen, enc, 'attention', 'attention+offsets', 0, 'cpu', np.pi,
                                             R, n)
            (-e2).backward()
            for pre, mod in (('e.', enc), ('d.', gen)):
                for k_, v in mod.named_parameters():
                    g0 = out['g' + pre + k_]
                    d = float((v.grad - g0).abs().max() / g0.abs().max().clamp_min(1e-30))
                    sens['k' + pre + k_] = max(sens.get('k' + pre + k_, 0.0), d)
        for k_, v in sens.items():
            out[k_] = np.float64(v)
        save(name, **out)
        print('   ', name, 'elbo', float(elbo), 'log_p', float(logp), 'kl', float(kl))
        print('    kink conditioning (1e-5 input perturbation):', {k_: '%.1e' % v for k_, v in sens.items()})


def gen_cli():
    """argparse surface of the four reference scripts (flags, defaults, choices) -> cli_flags.json."""
    import argparse
    import importlib
    import json

    class _Captured(Exception):
        pass

    out = {}
    sys.path.insert(0, REF)
    for name in ('train_mnist', 'train_particles', 'train_galaxy', 'train_dsprites'):
        mod = importlib.import_module(name)
        orig = argparse.ArgumentParser.parse_args

        def grab(self, *a, **k):
            raise _Captured(self)
        argparse.ArgumentParser.parse_args = grab
        try:
            mod.main()
        except _Captured as e:
            parser = e.args[0]
        finally:
            argparse.ArgumentParser.parse_args = orig
        flags = {}
        for act in parser._actions:
            if act.dest == 'help':
                continue
            flags[act.dest] = dict(flags=list(act.option_strings), default=act.default,
                                   choices=list(act.choices) if act.choices else None,
                                   type=getattr(act.type, '__name__', None), nargs0=(act.nargs == 0))
        out[name] = flags
    sys.path.pop(0)
    path = os.path.join(HERE, 'cli_flags.json')
    json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
    print('wrote', path, {k: len(v) for k, v in out.items()})


if __name__ == '__main__':
    which = sys.argv[1:] or ['bank', 'groupconv', 'encoder', 'decoder', 'steps', 'epoch', 'cli', 'particles_tail', 'get_latent', 'mrc', 'secondary', 'hotpath', 'trajectory', 'wide']
    which = [w for w in which if not w.startswith('hot_')]
    for w in which:
        {'bank': gen_bank, 'groupconv': gen_groupconv, 'encoder': gen_encoder, 'decoder': gen_decoder,
         'steps': gen_steps, 'epoch': gen_epoch, 'cli': gen_cli, 'particles_tail': gen_particles_tail, 'get_latent': gen_get_latent, 'mrc': gen_mrc, 'secondary': gen_secondary, 'hotpath': gen_hotpath, 'trajectory': gen_trajectory, 'wide': gen_wide}[w]()
