"""GPU parity of the drop-in `src.models` classes and the ELBO step against the golden fixtures generated from
the reference (tests/golden/make_goldens.py) and against the CPU oracle on fresh seeded inputs.
Tolerance: 1e-4 relative fp32 on outputs (BASELINE.json north_star), gradients within 1e-3 of max-norm with
LeakyReLU-kink outliers allowed (conftest.assert_grad_close)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_encoder_grads, assert_grad_close, load_golden, rel_err, seeded_models, tdict
from oracle import tvae_oracle as O

pytestmark = pytest.mark.gpu
OUT_TOL = 1e-4
GRAD_TOL = 1e-3


def dev():
    return torch.device('cuda:0')


@pytest.fixture(params=['f32', 'x6', 'h3'], autouse=True)
def gemm_mode(request):
    """Every module / step parity test runs in all three fp32-equivalent arithmetic modes of the matrix products: exact fp32
    MFMA, 'x6' (exactly split 3 x bf16 operands, six products) and 'h3' (2 x fp16 parts, three products, power-of-two tensor
    scales; launches without an h3 instance run x6) -- at the SAME tolerances."""
    from tvae import _lib
    with _lib.arithmetic(request.param):
        yield request.param


def build_encoder(fx, prefix):
    import src.models as M
    cfg = [int(v) for v in fx['cfg']]
    n, cin, zd, C, k, p, R, refine, normal = cfg[:9]
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, cin, zd, kernels_num=C, kernels_size=k, padding=p, groupconv=R, rot_refinement=bool(refine),
        theta_prior=float(fx['theta_prior']), normal_prior_over_r=bool(normal))
    enc.load_state_dict({k_: v for k_, v in tdict(fx, prefix).items()})
    return enc.to(dev())


def build_generator(fx, prefix, zd, hid, n_out, L, resid, fourier, sigma):
    import src.models as M
    gen = M.SpatialGenerator(zd, hid, n_out=n_out, num_layers=L, resid=bool(resid), fourier_expansion=bool(fourier),
                             sigma=sigma)
    gen.load_state_dict({k_: v for k_, v in tdict(fx, prefix).items()})
    return gen.to(dev())


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'bank_*.npz'))))
def test_trans_filter_golden(path):
    import src.models as M
    fx = load_golden(os.path.basename(path)[:-4])
    C, Cin, _, k, _ = fx['weight'].shape
    R = fx['bank'].shape[1]
    gc = M.GroupConv(Cin, C, k, output_rot_dim=R)
    gc.weight.data.copy_(torch.from_numpy(fx['weight']))
    gc = gc.to(dev())
    tw = gc.trans_filter(dev())
    assert tw.shape == fx['bank'].shape
    assert rel_err(tw, fx['bank']) < 1e-5
    (tw * torch.from_numpy(fx['gbank']).to(dev())).sum().backward()
    assert rel_err(gc.weight.grad, fx['gweight']) < 1e-5


@pytest.mark.parametrize('name', ['groupconv_fwd_bwd', 'groupconv_cin3_k9_R4'])
def test_groupconv_golden(name):
    import src.models as M
    fx = load_golden(name)
    C, Cin, _, k, _ = fx['weight'].shape
    R = fx['out'].shape[2]
    n, Ho = fx['y'].shape[-1], fx['out'].shape[-1]
    pad = (Ho - 1 + k - n) // 2
    gc = M.GroupConv(Cin, C, k, padding=pad, output_rot_dim=R)
    gc.weight.data.copy_(torch.from_numpy(fx['weight']))
    gc.bias.data.copy_(torch.from_numpy(fx['bias']))
    gc = gc.to(dev())
    out = gc(torch.from_numpy(fx['y']).to(dev()), dev())
    assert out.shape == fx['out'].shape
    assert rel_err(out, fx['out']) < OUT_TOL
    (out * torch.from_numpy(fx['gout']).to(dev())).sum().backward()
    assert rel_err(gc.weight.grad, fx['gweight']) < OUT_TOL
    assert rel_err(gc.bias.grad, fx['gbias']) < OUT_TOL


def test_groupconv_rot90_equivariance_gpu():
    """Known-answer property that needs no reference: 90-degree equivariance of the lifting conv."""
    import src.models as M
    torch.manual_seed(1)
    R = 8
    gc = M.GroupConv(1, 5, 12, padding=4, bias=False, output_rot_dim=R).to(dev())
    y = torch.rand(2, 1, 12, 12, device=dev())
    a = gc(torch.rot90(y, 1, dims=(2, 3)), dev())
    b = torch.rot90(torch.roll(gc(y, dev()), -R // 4, dims=2), 1, dims=(3, 4))
    assert rel_err(a, b) < 1e-5


@pytest.mark.parametrize('name', ['encoder_P8_28', 'encoder_P16_28_normal', 'encoder_P4_20_norefine',
                                  'encoder_P8_64'])
def test_encoder_golden(name):
    fx = load_golden(name)
    enc = build_encoder(fx, 'p.')
    outs = enc(torch.from_numpy(fx['y']).to(dev()), dev(), E=torch.from_numpy(fx['E']).to(dev()))
    attn, q, p_r, a_s, offs, theta, z = outs
    for got, key in ((attn, 'attn'), (q, 'q_t_r'), (p_r, 'p_r'), (a_s, 'a_sampled'), (offs, 'offsets'),
                     (theta, 'theta'), (z, 'z')):
        assert tuple(got.shape) == tuple(fx[key].shape), key
        assert rel_err(got, fx[key]) < OUT_TOL, key
    assert abs(float(torch.exp(q).reshape(q.shape[0], -1).sum(1).max()) - 1) < 1e-4
    w = {k_: torch.from_numpy(fx[k_]).to(dev()) for k_ in ('w_q', 'w_a', 'w_t', 'w_z')}
    probe = (q * w['w_q']).sum() + (a_s * w['w_a']).sum() * 50 + (theta * w['w_t']).sum() \
        + (z * w['w_z']).sum() + (attn * w['w_q']).sum() * 0.5
    probe.backward()
    assert_encoder_grads(enc, fx, GRAD_TOL, prefix='g.', name='', elbo_loss=False)


@pytest.mark.parametrize('actname', ['leakyrelu', 'tanh'])
def test_encoder_tail_paths_agree(actname):
    """128-channel encoder in the default arithmetic: the fused tail (conv2 + heads in one kernel; LeakyReLU: fused data
    gradient from the sign words; tanh: fused forward, unfused backward) against the unfused kernels on the same input --
    head outputs within 1e-5, every encoder gradient within 2e-4 (LeakyReLU rows with a flipped kink allowed as in
    assert_grad_close)."""
    import src.models as M
    from tvae import _lib, ops
    torch.manual_seed(4)
    act = torch.nn.LeakyReLU if actname == 'leakyrelu' else torch.nn.Tanh
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        28, 1, 2, kernels_num=128, kernels_size=28, padding=14, groupconv=8, rot_refinement=True, theta_prior=np.pi,
        normal_prior_over_r=False, activation=act).to(dev())
    y = torch.randn(6, 1, 28, 28, device=dev())
    w = torch.randn(7, 6 * 8 * 29 * 29, device=dev())

    def run(fused):
        for p in enc.parameters():
            p.grad = None
        old, ops.FUSE_ENC_TAIL, ops.PATH_LOG = ops.FUSE_ENC_TAIL, fused, set()
        try:
            with _lib.arithmetic('x6'):
                heads = enc.encode_heads(y)
            (heads * w).sum().backward()
            took = set(ops.PATH_LOG)
        finally:
            ops.FUSE_ENC_TAIL, ops.PATH_LOG = old, None
        return heads.detach().clone(), {k_: p.grad.clone() for k_, p in enc.named_parameters() if p.grad is not None}, took
    h1, g1, t1 = run(True)
    h0, g0, t0 = run(False)
    assert 'enc.tail_fwd_x6' in t1 and 'enc.tail_fwd_x6' not in t0
    assert ('enc.tail_dgrad_x6' in t1) == (actname == 'leakyrelu')
    assert rel_err(h1, h0) < 1e-5
    assert set(g1) == set(g0) and len(g1) >= 6
    for k_ in g0:
        assert_grad_close(g1[k_], g0[k_], tol=GRAD_TOL, name=k_)


@pytest.mark.parametrize('name', ['decoder_plain', 'decoder_plain512', 'decoder_fourier', 'decoder_resid',
                                  'decoder_nout2', 'decoder_nout3_z50_L4', 'decoder_z0_L1'])
def test_decoder_golden(name):
    fx = load_golden(name)
    zd, hid, n_out, L, resid, fourier = [int(v) for v in fx['cfg']]
    gen = build_generator(fx, 'p.', zd, hid, n_out, L, resid, fourier, float(fx['sigma']) if fourier else 0.01)
    x = torch.from_numpy(fx['x']).to(dev()).requires_grad_(True)
    z = torch.from_numpy(fx['z']).to(dev()).requires_grad_(True) if zd > 0 else None
    yh = gen(x, z)
    assert tuple(yh.shape) == tuple(fx['y_hat'].shape)
    assert rel_err(yh, fx['y_hat']) < OUT_TOL
    (yh * torch.from_numpy(fx['gy']).to(dev())).sum().backward()
    assert_grad_close(x.grad.reshape(-1, 2), fx['gx'].reshape(-1, 2), tol=GRAD_TOL, name='gx')
    if zd > 0:
        assert_grad_close(z.grad, fx['gz'], tol=GRAD_TOL, name='gz')
    for k_, t in gen.named_parameters():
        assert_grad_close(t.grad, fx['g.' + k_], tol=GRAD_TOL, name=k_)


STEP_LIK = {'step_mnist28_P8_init': 'bce', 'step_mnist28_P8_peaked': 'bce',
            'step_mnist28_P16_fourier_normal': 'bce', 'step_mnist28_P4_attention_resid': 'bce',
            'step_particles64_P8': 'gauss', 'step_particles32_fitnoise': 'gauss_var', 'step_galaxy_small': 'bce'}


def build_step_models(fx):
    n, cin, zd, C, k, p, R, refine, normal, hid, L, n_out, fourier, resid = [int(v) for v in fx['cfg']]
    enc = build_encoder(fx, 'e.')
    gen = build_generator(fx, 'd.', zd, hid, n_out, L, resid, fourier, float(fx['sigma']))
    return enc, gen, n


@pytest.mark.parametrize('name', sorted(STEP_LIK))
def test_step_golden(name):
    """eval_minibatch (reference train_*.py) -> (elbo, log_p, kl) and every parameter gradient."""
    from tvae import step
    fx = load_golden(name)
    enc, gen, n = build_step_models(fx)
    x = O.image_coords(n).to(dev())
    noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    elbo, logp, kl = step.elbo_terms(x, torch.from_numpy(fx['y']).to(dev()), gen, enc, STEP_LIK[name], noise)
    assert elbo.dtype == torch.float64 and kl.dtype == torch.float64 and logp.dtype == torch.float32
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < OUT_TOL
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < OUT_TOL
    assert abs(float(kl) - float(fx['kl'])) / abs(float(fx['kl'])) < OUT_TOL
    (-elbo).backward()
    assert_encoder_grads(enc, fx, GRAD_TOL)
    for k_, t in gen.named_parameters():
        assert_grad_close(t.grad, fx['gd.' + k_], tol=GRAD_TOL, name='gen.' + k_)


HOT = {'hot_S64_B2': ('gauss', {'conv1.dft', 'conv1.dft_ring', 'dec.virt_act', 'dec.fused_out', 'dec.virt_grad', 'dec.virt_grad_2val',
                                'dec.sign_bits', 'dec.fuse_in', 'dec.row_sums_in_dgrad', 'enc.tail_fwd_x6', 'enc.tail_dgrad_x6'}),
       'hot_S28F_B8': ('bce', {'conv1.dft', 'dec.four_x6', 'dec.fused_out', 'dec.virt_grad', 'dec.virt_grad_2val',
                               'dec.sign_bits', 'dec.row_sums_in_dgrad', 'enc.tail_fwd_x6', 'enc.tail_dgrad_x6'}),
       # the reference's real MNIST-U / MNIST-N geometry (50x50, k = 28, p = 8: train_mnist.py:413-417): the 66-wide frame
       # has its own instances of the transforms along w (ring kernels); 2 x 2500 pixels is not a multiple of 128, so the
       # decoder of this small batch takes the fp32-MFMA layers (a multiple of 32 images takes the split pipe)
       'hot_M50_B2': ('bce', {'conv1.dft', 'conv1.dft_ring', 'enc.tail_fwd_x6', 'enc.tail_dgrad_x6'})}


@pytest.mark.parametrize('name', sorted(HOT))
def test_step_hot_widths_golden(name, gemm_mode):
    """The branches bench.py times, against the real reference at FULL widths (hidden 512, C = 128): S64 (64x64, P8:
    recomputed first layer, fused output dot, implicit last-layer gradient, fused first-layer backward, frequency-domain
    convolution) and S28F (P16, Fourier first layer with [Wc | Wl] on the split pipe).  ELBO terms within 1e-4 and every
    parameter gradient within 1e-3 of max-norm; in the default arithmetic the test also asserts that those fused branches
    and the split-pipe GEMM entry points were the ones that ran."""
    from tvae import ops, step
    fx = load_golden(name)
    lik, want = HOT[name]
    enc, gen, n = seeded_models(fx)
    enc, gen = enc.to(dev()), gen.to(dev())
    x = O.image_coords(n).to(dev())
    noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    ops.PATH_LOG, ops.KERNEL_EVENTS, ops.PARTS_LOG = set(), {}, {}
    try:
        elbo, logp, kl = step.elbo_terms(x, torch.from_numpy(fx['y']).to(dev()), gen, enc, lik, noise)
        (-elbo).backward()
        torch.cuda.synchronize()
        took, events, parts_log = set(ops.PATH_LOG), set(ops.KERNEL_EVENTS), dict(ops.PARTS_LOG)
    finally:
        ops.PATH_LOG, ops.KERNEL_EVENTS, ops.PARTS_LOG = None, None, None
    if gemm_mode in ('x6', 'h3'):
        assert want <= took, (want - took, took)
        assert ({'tvae_conv1_fwd', 'tvae_conv1_wgrad'} if name == 'hot_M50_B2' else
                {'tvae_linear_fwd_x6', 'tvae_linear_dgrad_x6', 'tvae_linear_wgrad_x6', 'tvae_conv1_fwd',
                 'tvae_conv1_wgrad'}) <= events, events
    if name == 'hot_S64_B2' and gemm_mode in ('x6', 'h3'):
        # the arithmetic that REACHED the entry points: in h3 every big launch of the timed step runs
        # its two-part instance (parts == 2; 3 / 2 matrix instructions per product block), in x6 the three-part one
        p = 2 if gemm_mode == 'h3' else 3
        blocks = {'tvae_conv1_fwd': ops.mfma_per_block(p), 'tvae_conv1_wgrad': ops.mfma_per_block(p),
                  'tvae_linear_fwd_x6': ops.mfma_per_block(p), 'tvae_linear_dgrad_x6': ops.mfma_per_block(p, True),
                  'tvae_linear_wgrad_x6': ops.mfma_per_block(p, True), 'tvae_enc_tail_fwd_x6': ops.mfma_per_block(p),
                  'tvae_enc_tail_dgrad_x6': ops.mfma_per_block(p)}     # (conv2's fused weight gradient needs N % 32 == 0: bench-size test)
        for k_, b_ in blocks.items():
            assert parts_log.get(k_) == [(p, b_)], (k_, parts_log.get(k_))
    if name == 'hot_S28F_B8' and gemm_mode == 'h3':
        # the h3 routes of operands streamed from MEMORY (Fourier first layer, the hidden layer behind it, the
        # first layer's weight gradient and its data gradient under caller-supplied bounds) must not fall back silently
        assert {'dec.four_h3', 'dec.hidden_h3_mem'} <= took, took
        assert all(p_ == 2 for p_, _ in parts_log['tvae_linear_fwd_x6']), parts_log['tvae_linear_fwd_x6']
        assert all(p_ == 2 for p_, _ in parts_log['tvae_linear_wgrad_x6']), parts_log['tvae_linear_wgrad_x6']
        assert all(p_ == 2 for p_, _ in parts_log['tvae_linear_dgrad_x6']), parts_log['tvae_linear_dgrad_x6']
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < OUT_TOL
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < OUT_TOL
    assert abs(float(kl) - float(fx['kl'])) / abs(float(fx['kl'])) < OUT_TOL
    # gradient gate: 1e-3 of max-norm, or twice the fixture's own kink conditioning where that is larger -- `ke.*` /
    # `kd.*` is how far the REFERENCE's gradient moves under a 1e-5 relative input perturbation (a few LeakyReLU
    # pre-activations out of 3-6 M cross 0; make_goldens.py:gen_hotpath); no free outlier for single-row tensors
    assert_encoder_grads(enc, fx, GRAD_TOL, kink_prefix='ke.')
    for k_, t in gen.named_parameters():
        assert_grad_close(t.grad, fx['gd.' + k_], tol=max(GRAD_TOL, 2 * float(fx['kd.' + k_])), name='gen.' + k_)


@pytest.mark.parametrize('cfg', ['small', 'small_fourier', 'S28', 'S28F', 'S64', 'S64x', 'M28', 'M28r', 'M50', 'M50x', 'G96'])
def test_step_does_not_read_out_of_bounds(cfg):
    """Out-of-bounds READ detector.  Every float tensor the step allocates (torch.empty / torch.zeros, workspaces
    included) becomes a view into the middle of a larger allocation; ELBO and every gradient must be bitwise independent
    of what the guard bands hold (0, 1e4, NaN) -- a kernel that reads past a row end or a padded tile and lets it into
    a result with any weight fails here.  The INTERIOR of every torch.empty allocation (scratch / workspace tensors
    included) starts out with the same fill value, so an element that is consumed before any kernel wrote it (an
    unwritten workspace slab, an accumulate-into-uninitialised launch) changes the result between the three runs too.
    Full kernel widths (C = 128, hidden 512) at small batches."""
    import src.models as M
    from tvae import ops, step, tables
    n, zd, R, B, C, hid, k, pad, four = {'small': (20, 2, 8, 8, 8, 32, 20, 4, False),
                                         'small_fourier': (20, 2, 8, 8, 8, 32, 20, 4, True),
                                         'S28': (28, 2, 8, 16, 128, 512, 28, 14, False),
                                         'S28F': (28, 2, 16, 8, 128, 512, 28, 14, True),
                                         'S64': (64, 2, 8, 4, 128, 512, 64, 16, False),
                                         # B * Ho a multiple of 32 (as at the bench's B = 256): no ragged last tile
                                         'S64x': (64, 2, 8, 32, 128, 512, 64, 16, False),
                                         'M50x': (50, 2, 8, 32, 128, 512, 28, 8, False),
                                         # the reference's MNIST geometries (k = 28, padding 8): the 44- and 66-wide ring
                                         # transforms; M28r: a batch whose last 32-column tile is ragged
                                         'M28': (28, 2, 8, 32, 128, 512, 28, 8, False),
                                         'M28r': (28, 2, 8, 5, 128, 512, 28, 8, False),
                                         'M50': (50, 2, 8, 4, 128, 512, 28, 8, False),
                                         # round 5: a large frame (L = 112, Ho = 97 = 3 x 32 + 1): the WIDE generic transforms
                                         # along w (workgroup per tile, extra output column on the vector ALU), ragged batch
                                         'G96': (96, 2, 8, 3, 16, 64, 32, 16, False)}[cfg]
    torch.manual_seed(0)
    gen = M.SpatialGenerator(zd, hid, num_layers=2, fourier_expansion=four, sigma=2.0 / (n - 1)).to(dev())
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, 1, zd, kernels_num=C, kernels_size=k, padding=pad, groupconv=R, rot_refinement=True, theta_prior=np.pi,
        normal_prior_over_r=False).to(dev())
    x = torch.from_numpy(tables.image_coords(n)).to(dev())
    y = torch.randn(B, 1, n, n, device=dev())
    nz = step.draw_noise(B, R * enc.output_size() ** 2, zd, dev())
    params = list(gen.named_parameters()) + list(enc.named_parameters())
    real_empty, real_zeros = torch.empty, torch.zeros
    guard, fill = 8192, [0.0]

    def guarded(real):
        def f(*a, **kw):
            t = real(*a, **kw)
            if not (t.is_cuda and t.is_floating_point() and t.dim() >= 1 and t.numel() > 0):
                return t
            big = real_empty(t.numel() + 2 * guard, dtype=t.dtype, device=t.device)
            big[:guard].fill_(fill[0])
            big[guard + t.numel():].fill_(fill[0])
            v = big[guard:guard + t.numel()].view(t.shape)
            if real is real_zeros:
                v.zero_()
            else:
                v.fill_(fill[0])      # torch.empty interior: an element consumed before any kernel wrote it shows too
            return v
        return f

    def run(g):
        fill[0] = g
        saved = dict(ops._WS)
        ops._WS.clear()                                  # workspaces are re-created under guard too
        torch.empty, torch.zeros = guarded(real_empty), guarded(real_zeros)
        try:
            for _, p in params:
                p.grad = None
            e, _, _ = step.elbo_terms(x, y, gen, enc, 'gauss', nz)
            (-e).backward()
            return float(e.detach()), {k_: p.grad.clone() for k_, p in params}
        finally:
            torch.empty, torch.zeros = real_empty, real_zeros
            ops._WS.clear()
            ops._WS.update(saved)
    e0, g0 = run(0.0)
    for g in (1e4, float('nan')):
        e1, g1 = run(g)
        assert e1 == e0, (g, e0, e1)
        for k_ in g0:
            assert torch.equal(g0[k_], g1[k_]), (g, k_)


@pytest.mark.parametrize('name', ['hot_S28F_B8', 'hot_S64_B2'])
def test_bf16_throughput_mode(name):
    """The opt-in bf16 throughput mode (operands rounded to ONE bf16 number, one MFMA per product block, fp32
    accumulate; BASELINE.json configs 2 / 5) on the same full-width steps and the same fused branches.  It is NOT
    fp32-equivalent and is held to its own, stated tolerance: ELBO terms within 2e-2 relative of the reference, the
    gradient of every parameter tensor within 25 % of its max-norm and at a cosine of >= 0.98 to the reference's."""
    from tvae import _lib, ops, step
    fx = load_golden(name)
    lik, want = HOT[name]
    enc, gen, n = seeded_models(fx)
    enc, gen = enc.to(dev()), gen.to(dev())
    x = O.image_coords(n).to(dev())
    noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    ops.PATH_LOG = set()
    try:
        with _lib.arithmetic('bf16'):
            elbo, logp, kl = step.elbo_terms(x, torch.from_numpy(fx['y']).to(dev()), gen, enc, lik, noise)
        (-elbo).backward()                  # outside the block: the backward runs in its forward's arithmetic
        torch.cuda.synchronize()
        took = set(ops.PATH_LOG)
    finally:
        ops.PATH_LOG = None
    assert want <= took, (want - took, took)
    for got, key in ((elbo, 'elbo'), (logp, 'log_p'), (kl, 'kl')):
        assert abs(float(got) - float(fx[key])) / abs(float(fx[key])) < 2e-2, (key, float(got), float(fx[key]))
    for prefix, mod in (('ge.', enc), ('gd.', gen)):
        for k_, t in mod.named_parameters():
            if k_ == 'conv_a.bias':
                continue                    # analytically zero
            a, b = t.grad.double().cpu().reshape(-1), torch.from_numpy(fx[prefix + k_]).double().reshape(-1)
            assert float((a - b).abs().max() / b.abs().max()) < 0.25, (k_, float((a - b).abs().max() / b.abs().max()))
            assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.98, k_


@pytest.mark.parametrize('name,lik', [('step_mnist28_P8_init', 'bce'), ('step_galaxy_small', 'bce')])
def test_bf16_throughput_mode_named_configs(name, lik):
    """The bf16 mode on the configurations BASELINE.json names it for: configs[1] (MNIST-U 28x28, P8, z = 2: the full-width
    reference step fixture) and configs[4] (galaxy: 3 channels, P16, z = 50, Fourier decoder, 4 layers, 3 outputs -- at the
    small fixture's sizes).  Own, stated tolerance as in test_bf16_throughput_mode: ELBO terms within 2e-2 of the reference,
    every gradient tensor within 25 % of its max-norm at a cosine of >= 0.98 (galaxy fixture, whose layers are 16 / 64 wide
    and therefore run on the fp32-MFMA kernels outside the convolution: 0.97)."""
    from tvae import _lib, step
    fx = load_golden(name)
    enc, gen, n = build_step_models(fx)
    x = O.image_coords(n).to(dev())
    noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    with _lib.arithmetic('bf16'):
        elbo, logp, kl = step.elbo_terms(x, torch.from_numpy(fx['y']).to(dev()), gen, enc, lik, noise)
    (-elbo).backward()
    torch.cuda.synchronize()
    for got, key in ((elbo, 'elbo'), (logp, 'log_p'), (kl, 'kl')):
        assert abs(float(got) - float(fx[key])) / abs(float(fx[key])) < 2e-2, (key, float(got), float(fx[key]))
    cos_min = 0.97 if name == 'step_galaxy_small' else 0.98
    for prefix, mod in (('ge.', enc), ('gd.', gen)):
        for k_, t in mod.named_parameters():
            if k_ == 'conv_a.bias':
                continue                    # analytically zero
            a, b = t.grad.double().cpu().reshape(-1), torch.from_numpy(fx[prefix + k_]).double().reshape(-1)
            assert float((a - b).abs().max() / b.abs().max()) < 0.25, (k_, float((a - b).abs().max() / b.abs().max()))
            assert float(torch.dot(a, b) / (a.norm() * b.norm())) > cos_min, k_


def test_trajectory_20_steps_golden(gemm_mode):
    """20 consecutive reference Adam steps (train_mnist.py:300-346; lr 2e-3, 4 images per step): the per-step ELBO / Error /
    KL curve within 1e-4 (the KL term, 2 % of the ELBO and the part that moves with the attention, within 1e-3: measured 2.6e-4
    at step 14) and the 20-step parameter UPDATE within 2e-2 relative L2 / cosine 0.999 per tensor
    (conftest.assert_trajectory_update_close: Adam turns rounding-level gradient entries into +-lr steps, so two fp32
    evaluations -- the reference and its CPU restatement already differ by up to 9.4e-3 -- cannot agree element-wise).

    Along the way every step's gradient in the split arithmetics (x6, h3) is compared with the exact-fp32 gradient AT THE SAME
    PARAMETERS: 3e-5 per tensor (measured: <= 6.4e-6) -- unless a LeakyReLU output changed sign between the two evaluations (an element whose
    pre-activation is zero to rounding: which side it falls on is decided by the last bit in ANY arithmetic, and its mask
    then differs by a factor 100).  Such a kink flip is counted, must be visible as a sign difference in a recorded
    activation, and widens the final-update gate to 1e-1 / cosine 0.995: measured, ONE flipped element of the 200 704 in the
    decoder's last hidden layer at step 1 (values +7e-9 / -5e-10) changes that step's decoder gradients by 3e-4 and the
    20-step update by 3-4.5e-2 (h3; the x6 run happens to flip none and stays at 1.1e-2)."""
    from conftest import assert_trajectory_update_close
    from tvae import _lib, optim, step, ops
    fx = load_golden('trajectory_20steps')
    enc, gen, n = build_step_models(fx)
    params = list(gen.parameters()) + list(enc.parameters())
    names = ['d.' + k_ for k_, _ in gen.named_parameters()] + ['e.' + k_ for k_, _ in enc.named_parameters()]
    opt = optim.FlatAdam(params, lr=float(fx['lr']))
    data = torch.from_numpy(fx['data']).to(dev())
    x = O.image_coords(n).to(dev())
    T, B = fx['curve'].shape[0], fx['E'].shape[1]
    # outputs of the launches that apply LeakyReLU in this configuration (position of the output in the argument list)
    out_arg = {'tvae_dec_l0_fwd': 4, 'tvae_linear_fwd': 6, 'tvae_conv1_fwd_dft': 3, 'tvae_conv1_fwd': 3, 'tvae_conv1_fwd_x6': 3}
    acts, real_call = [], ops.call

    def recording_call(name, *args):
        r = real_call(name, *args)
        if name in out_arg:
            acts.append(args[out_arg[name]].detach().clone())
        return r
    flips = 0
    for t in range(T):
        nz = tuple(torch.from_numpy(fx[k_][t]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
        yb = data[B * t:B * t + B]
        if gemm_mode != 'f32':
            grads, signs = {}, {}
            ops.call = recording_call
            try:
                for mode in ('f32', gemm_mode):
                    acts.clear()
                    opt.zero_grad()
                    with _lib.arithmetic(mode):
                        e_, _, _ = step.elbo_terms(x, yb, gen, enc, 'bce', nz)
                        (-e_).backward()
                    grads[mode] = [p.grad.detach().clone().double() for p in params]
                    signs[mode] = [a > 0 for a in acts]
            finally:
                ops.call = real_call
            opt.zero_grad()
            bad = [(nm, float((g1 - g0).norm() / g0.norm())) for nm, g0, g1 in zip(names, grads['f32'], grads[gemm_mode])
                   if nm != 'e.conv_a.bias' and float((g1 - g0).norm()) > 3e-5 * float(g0.norm())]
            if bad:
                nflip = sum(int((a != b).sum()) for a, b in zip(signs['f32'], signs[gemm_mode]))
                assert len(signs['f32']) == len(signs[gemm_mode]) and 1 <= nflip <= 4, (t, bad, nflip)
                # (round 5: the shorter circular frame changed the rounding of the convolution and with it WHICH elements sit
                #  on a kink: step 18 now flips one whose effect on d.layers.1.bias is 3.4e-3 -- a flip moves a bias gradient of
                #  this 4-image fixture by (1 - slope) wo gy of ONE element against a sum over 3 136)
                assert max(v for _, v in bad) < 6e-3, (t, bad)
                flips += nflip
        e, err, kl = step.train_epoch([(yb,)], x, gen, enc, opt, 'attention', 'attention+offsets', 0, 1,
                                      B, dev(), params, np.pi, 8, n, progress=False, noise_iter=iter([nz]))
        want = fx['curve'][t]
        assert abs(e - want[0]) / abs(want[0]) < OUT_TOL and abs(err - want[1]) / abs(want[1]) < OUT_TOL, (t, e, err, want)
        assert abs(kl - want[2]) / abs(want[2]) < 1e-3, (t, kl, want)
    if flips:
        assert_trajectory_update_close(enc.state_dict(), gen.state_dict(), fx, 1e-1, cos_min=0.995)
    else:
        assert_trajectory_update_close(enc.state_dict(), gen.state_dict(), fx, 2e-2)


def test_particles_tail_wide_golden(gemm_mode):
    """CTF + circular mask likelihood tail (train_particles.py:298-338) at the reference's DEFAULT widths: 128 kernels, hidden
    512 (seed-based fixture, parameters re-created and checked against digests)."""
    from tvae import ops, step
    fx = load_golden('wide_particles32_ctf_mask')
    enc, gen, n = seeded_models(fx)
    enc, gen = enc.to(dev()), gen.to(dev())
    x = O.image_coords(n).to(dev())
    noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    ops.PATH_LOG = set()
    try:
        elbo, logp, kl = step.eval_minibatch_particles(x, torch.from_numpy(fx['y']).to(dev()),
                                                       torch.from_numpy(fx['ctf']).to(dev()), gen, enc, 'attention',
                                                       'attention+offsets', 0, dev(), np.pi, 8, 8, int(fx['mask_radius']),
                                                       noise=noise)
        (-elbo).backward()
        took = set(ops.PATH_LOG)
    finally:
        ops.PATH_LOG = None
    if gemm_mode in ('x6', 'h3'):
        assert {'conv1.dft', 'enc.tail_fwd_x6', 'enc.tail_dgrad_x6'} <= took, took
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < OUT_TOL
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < OUT_TOL
    assert abs(float(kl) - float(fx['kl'])) / abs(float(fx['kl'])) < OUT_TOL
    assert_encoder_grads(enc, fx, GRAD_TOL)
    for k_, t in gen.named_parameters():
        assert_grad_close(t.grad, fx['gd.' + k_], tol=GRAD_TOL, name='gen.' + k_)


def test_secondary_branch_wide_golden(gemm_mode):
    """--r-inf unimodal with groupconv 4 (src/models.py:268-319, train_mnist.py:86-185) at the reference's default widths:
    128 kernels through the lifting convolution, fc_r rotation pooling, conv2 and the heads; decoder hidden 512."""
    import src.models as M
    from tvae import ops, step
    fx = load_golden('wide_attention_unimodal_gc4')
    n, zd, gc, C, hid = [int(v) for v in fx['cfg']]
    torch.manual_seed(int(fx['seed']))
    gen = M.SpatialGenerator(zd, hid, num_layers=2)
    enc = M.InferenceNetwork_AttentionTranslation_UnimodalRotation(n, 1, zd, kernels_num=C, groupconv=gc)
    with torch.no_grad():
        for nm in ('conv_a', 'conv_r', 'conv_z'):
            getattr(enc, nm).weight.mul_(float(fx['scale_heads']))
    for prefix, mod in (('se.', enc), ('sd.', gen)):             # seeded construction == the reference's (digests)
        for k_, v in mod.state_dict().items():
            t = v.detach().double().reshape(-1)
            want = fx[prefix + k_]
            assert float(t[0]) == want[2] and float(t[-1]) == want[3] and abs(float(t.abs().sum()) - want[1]) <= 1e-11 * want[1], k_
    enc, gen = enc.to(dev()), gen.to(dev())
    noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    ops.PATH_LOG = set()
    try:
        elbo, logp, kl = step.eval_minibatch(O.image_coords(n).to(dev()), torch.from_numpy(fx['y']).to(dev()), gen, enc,
                                             'attention', 'unimodal', 0, dev(), np.pi, gc, n, noise=noise)
        taken = set(ops.PATH_LOG)
    finally:
        ops.PATH_LOG = None
    assert 'trans_attn.rot_pool' in taken, taken
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < OUT_TOL
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < OUT_TOL
    assert abs(float(kl) - float(fx['kl'])) / abs(float(fx['kl'])) < OUT_TOL
    (-elbo).backward()
    assert_encoder_grads(enc, fx, GRAD_TOL)
    for k_, t in gen.named_parameters():
        assert_grad_close(t.grad, fx['gd.' + k_], tol=GRAD_TOL, name='gen.' + k_)


def test_step_hot_widths_intermediates_vs_oracle():
    """Reconstruction, latent sample, theta, dx of the S64-width step against the pinned oracle (default arithmetic)."""
    from tvae import step
    fx = load_golden('hot_S64_B2')
    enc, gen, n = seeded_models(fx)
    cfgv = [int(v) for v in fx['cfg']]
    noise_c = dict(E=torch.from_numpy(fx['E']), eps_z=torch.from_numpy(fx['eps_z']),
                   eps_theta=torch.from_numpy(fx['eps_theta']))
    with torch.no_grad():
        _, _, _, aux = O.elbo_step(O.image_coords(n), torch.from_numpy(fx['y']), enc.state_dict(), gen.state_dict(),
                                   R=cfgv[6], padding=cfgv[5], rot_refinement=True, theta_prior=float(fx['theta_prior']),
                                   normal_prior_over_r=False, num_layers=cfgv[10], resid=False, fourier_sigma=None,
                                   likelihood='gauss', return_aux=True, **noise_c)
    enc, gen = enc.to(dev()), gen.to(dev())
    noise = tuple(t.to(dev()) for t in noise_c.values())
    with torch.no_grad():
        _, _, _, got = step.elbo_terms(O.image_coords(n).to(dev()), torch.from_numpy(fx['y']).to(dev()), gen, enc,
                                       'gauss', noise, return_aux=True)
    for k_ in ('z', 'theta', 'dx', 'x_rot', 'y_hat', 'kl_per_image', 'a_sampled', 'q_t_r'):
        assert rel_err(got[k_].reshape(-1), aux[k_].reshape(-1)) < OUT_TOL, k_


def test_step_intermediates_vs_oracle():
    """Latent sample z, theta, dx, reconstruction y_hat against the oracle on the peaked fixture (SURVEY 8d gate)."""
    from tvae import step
    fx = load_golden('step_mnist28_P8_peaked')
    enc, gen, n = build_step_models(fx)
    cfgv = [int(v) for v in fx['cfg']]
    noise_c = dict(E=torch.from_numpy(fx['E']), eps_z=torch.from_numpy(fx['eps_z']),
                   eps_theta=torch.from_numpy(fx['eps_theta']))
    _, _, _, aux = O.elbo_step(O.image_coords(n), torch.from_numpy(fx['y']), tdict(fx, 'e.'), tdict(fx, 'd.'),
                               R=cfgv[6], padding=cfgv[5], rot_refinement=bool(cfgv[7]),
                               theta_prior=float(fx['theta_prior']), normal_prior_over_r=bool(cfgv[8]),
                               num_layers=cfgv[10], resid=bool(cfgv[13]), fourier_sigma=None, likelihood='bce',
                               return_aux=True, **noise_c)
    x = O.image_coords(n).to(dev())
    noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    with torch.no_grad():
        _, _, _, got = step.elbo_terms(x, torch.from_numpy(fx['y']).to(dev()), gen, enc, 'bce', noise,
                                       return_aux=True)
    assert float(aux['a_sampled'].max()) > 0.2          # attention really is peaked in this fixture
    for k_ in ('z', 'theta', 'dx', 'x_rot', 'y_hat', 'kl_per_image', 'a_sampled', 'q_t_r'):
        assert rel_err(got[k_].reshape(-1), aux[k_].reshape(-1)) < OUT_TOL, k_


def test_epoch_two_steps_golden(gemm_mode):
    """train_epoch with the fused flat Adam reproduces the reference running means and post-step parameters."""
    from tvae import optim, step
    fx = load_golden('epoch_2steps')
    enc, gen, n = build_step_models({**fx, **{k_: v for k_, v in fx.items()}})
    params = list(gen.parameters()) + list(enc.parameters())
    opt = optim.FlatAdam(params, lr=2e-4)
    data = torch.from_numpy(fx['data']).to(dev())
    noises = iter([tuple(torch.from_numpy(fx[f'{k_}{i}']).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
                   for i in range(2)])
    it = [(data[0:4],), (data[4:8],)]
    x = O.image_coords(n).to(dev())
    e, err, kl = step.train_epoch(it, x, gen, enc, opt, 'attention', 'attention+offsets', 0, 1, 8, dev(), params,
                                  np.pi, 8, n, progress=False, noise_iter=noises)
    assert abs(e - float(fx['elbo'])) / abs(float(fx['elbo'])) < OUT_TOL
    assert abs(err - float(fx['err'])) / abs(float(fx['err'])) < OUT_TOL
    assert abs(kl - float(fx['kl'])) / abs(float(fx['kl'])) < OUT_TOL
    for k_, t in enc.state_dict().items():
        if k_ == 'conv_a.bias':      # analytic gradient is 0 (softmax shift invariance): Adam follows rounding noise
            assert (t.cpu() - torch.from_numpy(fx['e1.' + k_])).abs().max() <= 2 * 2e-4 * 2 + 1e-7
            continue
        assert rel_err(t, fx['e1.' + k_]) < 2e-5, k_
    for k_, t in gen.state_dict().items():
        assert rel_err(t, fx['d1.' + k_]) < 2e-5, k_


def test_full_size_properties():
    """BASELINE.json metric configuration at FULL size (64x64, P8, z=2, k=64 p=16, C=128, hidden 512, B=256) --
    size-independent properties:
    exp(q) sums to 1, a sums to 1, KL >= 0 finite, determinism (bitwise equal on a second run), and the encoder's
    batch independence (image b's outputs do not depend on its neighbours)."""
    import src.models as M
    from tvae import step
    torch.manual_seed(0)
    n, R, B = 64, 8, 256
    gen = M.SpatialGenerator(2, 512, num_layers=2).to(dev())
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, 1, 2, kernels_num=128, kernels_size=64, padding=16, groupconv=R, rot_refinement=True,
        theta_prior=np.pi, normal_prior_over_r=False).to(dev())
    y = torch.randn(B, 1, n, n, device=dev())
    x = O.image_coords(n).to(dev())
    noise = step.draw_noise(B, R * 33 * 33, 2, dev())
    with torch.no_grad():
        e1, lp1, kl1, aux = step.elbo_terms(x, y, gen, enc, 'gauss', noise, return_aux=True)
        e2, lp2, kl2 = step.elbo_terms(x, y, gen, enc, 'gauss', noise)
        assert float(e1) == float(e2) and float(lp1) == float(lp2) and float(kl1) == float(kl2)
        assert (torch.exp(aux['q_t_r']).sum(1) - 1).abs().max() < 1e-4
        assert (aux['a_sampled'].sum(1) - 1).abs().max() < 1e-4
        assert torch.isfinite(aux['kl_per_image']).all() and (aux['kl_per_image'] > -1e-4).all()
        sub = slice(5, 9)
        _, _, _, aux_s = step.elbo_terms(x, y[sub], gen, enc, 'gauss', tuple(t[sub] for t in noise), return_aux=True)
        assert rel_err(aux_s['y_hat'], aux['y_hat'][sub]) < 1e-5
        assert rel_err(aux_s['kl_per_image'], aux['kl_per_image'][sub]) < 1e-5
    # gradients at the full size (every CU busy, persistent kernels with several workgroups per CU): two backward passes
    # are BITWISE identical (no atomics, fixed reduction orders -- a sporadic hazard would show here), and the default
    # arithmetic agrees with exact fp32 products within max(1e-3, 2 x the step's own conditioning under a 1e-6 relative
    # perturbation of the input)
    from tvae import _lib
    params = list(enc.named_parameters()) + list(gen.named_parameters())

    def grads(yy, mode):
        for _, p in params:
            p.grad = None
        with _lib.arithmetic(mode):
            e_, _, _ = step.elbo_terms(x, yy, gen, enc, 'gauss', noise)
        (-e_).backward()
        return {nm: p.grad.clone() for nm, p in params}
    # bitwise-reproducible gradients in BOTH split-pipe arithmetics: the shipped default (h3: its operand maxima come from
    # atomic maxima on bit patterns -- order independent -- and fire-and-forget slot updates) and the exact split
    for mode_ in ('h3', 'x6'):
        g1, g2 = grads(y, mode_), grads(y, mode_)
        for nm, _ in params:
            assert torch.equal(g1[nm], g2[nm]), (mode_, nm)
    gf = grads(y, 'f32')
    gp = grads(y * (1 + 1e-6 * torch.randn_like(y)), 'f32')
    for nm, _ in params:
        if nm == 'conv_a.bias':                          # analytically zero (softmax shift invariance): noise over noise
            continue
        scale = float(gf[nm].abs().max())
        cond = float((gp[nm] - gf[nm]).abs().max()) / scale
        diff = float((g1[nm] - gf[nm]).abs().max()) / scale
        assert diff < max(1e-3, 2 * cond), (nm, diff, cond)


def test_galaxy_full_size_runs(gemm_mode):
    """BASELINE configs[4] at full size (128x128x3, k=64 p=32, P16, z=50, Fourier, 4 decoder layers, n_out=3), B=2.
    In the default arithmetic the lifting convolution runs through the frequency domain (3 channels in the reduction,
    192-wide frame: blocked spectra, generic transforms along w) and the attention head through the chunked kernels; with
    exact fp32 products the padded image does not fit LDS and conv1 takes the generic implicit-GEMM loaders.
    Size-independent properties: finite ELBO, exp(q) and the Gumbel sample sum to 1, determinism, finite gradients for
    every parameter -- and the two arithmetics agree on the ELBO terms (1e-4) and on every gradient (1e-3 of max-norm or twice the step's own measured conditioning)."""
    import src.models as M
    from tvae import _lib, ops, step
    torch.manual_seed(0)
    n, R, B, zd = 128, 16, 2, 50
    gen = M.SpatialGenerator(zd, 512, n_out=3, num_layers=4, fourier_expansion=True, sigma=2.0 / (n - 1)).to(dev())
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, 3, zd, kernels_num=128, kernels_size=64, padding=32, groupconv=R, rot_refinement=True,
        theta_prior=np.pi, normal_prior_over_r=False).to(dev())
    Ho = n + 2 * 32 - 64 + 1
    y = torch.rand(B, 3, n, n, device=dev())
    x = O.image_coords(n).to(dev())
    noise = step.draw_noise(B, R * Ho * Ho, zd, dev())
    ops.PATH_LOG = set()
    try:
        e1, lp1, kl1, aux = step.elbo_terms(x, y, gen, enc, 'bce3', noise, return_aux=True)
        took = set(ops.PATH_LOG)
    finally:
        ops.PATH_LOG = None
    assert ('conv1.dft' in took) == (gemm_mode in ('x6', 'h3')), took
    assert torch.isfinite(e1) and torch.isfinite(lp1) and torch.isfinite(kl1)
    assert (torch.exp(aux['q_t_r']).sum(1) - 1).abs().max() < 2e-4
    assert (aux['a_sampled'].sum(1) - 1).abs().max() < 2e-4
    (-e1).backward()
    params = list(enc.named_parameters()) + list(gen.named_parameters())
    for nm, p in params:
        assert p.grad is not None and torch.isfinite(p.grad).all(), nm
    with torch.no_grad():
        e2, _, _ = step.elbo_terms(x, y, gen, enc, 'bce3', noise)
    assert float(e1) == float(e2)
    if gemm_mode in ('x6', 'h3'):                   # the same step with exact fp32 products
        def grads_f32(yy):
            for _, p in params:
                p.grad = None
            with _lib.arithmetic('f32'):
                e_, lp_, kl_ = step.elbo_terms(x, yy, gen, enc, 'bce3', noise)
            (-e_).backward()
            return (e_, lp_, kl_), {nm: p.grad.clone() for nm, p in params}
        g_x6 = {nm: p.grad.clone() for nm, p in params}
        (e3, lp3, kl3), g_f32 = grads_f32(y)
        for got, want in ((e1, e3), (lp1, lp3), (kl1, kl3)):
            assert abs(float(got) - float(want)) / abs(float(want)) < OUT_TOL
        # Gradient gate = max(1e-3, 2 x this step's own conditioning): 2 x 68 M encoder pre-activations pass a LeakyReLU
        # and a 1e-6 relative perturbation of the INPUT, inside one arithmetic, already moves every gradient by 2-4e-3
        # of its max-norm (kink flips; profiles/tools/diag_galaxy.py) -- as much as the two arithmetics differ.
        _, g_pert = grads_f32(y * (1 + 1e-6 * torch.randn_like(y)))
        for nm, _ in params:
            if nm == 'conv_a.bias':
                continue
            scale = float(g_f32[nm].abs().max())
            cond = float((g_pert[nm] - g_f32[nm]).abs().max()) / scale
            diff = float((g_x6[nm] - g_f32[nm]).abs().max()) / scale
            assert diff < max(1e-3, 2 * cond), (nm, diff, cond)


@pytest.mark.parametrize('name', ['step_particles32_ctf', 'step_particles32_mask', 'step_particles32_ctf_mask'])
def test_particles_tail_golden(name):
    """CTF filter + circular mask likelihood tail (train_particles.py:298-338) through eval_minibatch_particles."""
    from tvae import step
    fx = load_golden(name)
    enc, gen, n = build_step_models(fx)
    x = O.image_coords(n).to(dev())
    noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    ctf = torch.from_numpy(fx['ctf']).to(dev()) if 'ctf' in fx else None
    elbo, logp, kl = step.eval_minibatch_particles(x, torch.from_numpy(fx['y']).to(dev()), ctf, gen, enc, 'attention',
                                                   'attention+offsets', 0, dev(), np.pi, 8, 8,
                                                   int(fx['mask_radius']), noise=noise)
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < OUT_TOL
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < OUT_TOL
    (-elbo).backward()
    assert_encoder_grads(enc, fx, GRAD_TOL)
    for k_, t in gen.named_parameters():
        assert_grad_close(t.grad, fx['gd.' + k_], tol=GRAD_TOL, name='gen.' + k_)


@pytest.mark.parametrize('name', ['get_latent_P8_28', 'get_latent_P4_20_norefine'])
def test_get_latent_golden(name):
    """tvae.latent.get_latent against the reference's clustering_mnist.get_latent (golden from the real function)."""
    from tvae import latent
    fx = load_golden(name)
    enc = build_encoder(fx, 'p.')
    n = int(fx['cfg'][0])
    r_inf = 'attention+offsets' if int(fx['cfg'][7]) else 'attention'
    zc, th, dx = latent.get_latent(O.image_coords(n).to(dev()), torch.from_numpy(fx['y']).to(dev()), enc, 'attention',
                                   r_inf, dev(), n)
    assert tuple(zc.shape) == tuple(fx['z_content'].shape) and tuple(th.shape) == tuple(fx['theta_mu'].shape)
    assert rel_err(zc, fx['z_content']) < OUT_TOL
    assert rel_err(th, fx['theta_mu']) < OUT_TOL
    assert rel_err(dx, fx['dx']) < OUT_TOL


@pytest.mark.parametrize('resid,layers,act', [(False, 1, 'leakyrelu'), (True, 3, 'leakyrelu'), (False, 2, 'tanh')])
def test_mlp_encoder_matches_torch_stack(resid, layers, act):
    """MLP encoder (reference models.py:229-260) on the GEMM kernels against the same nn.Sequential evaluated by torch."""
    import src.models as M
    from tvae import ops
    torch.manual_seed(3)
    A = torch.nn.LeakyReLU if act == 'leakyrelu' else torch.nn.Tanh
    enc = M.InferenceNetwork_UnimodalTranslation_UnimodalRotation(49, 5, 40, num_layers=layers, activation=A,
                                                                  resid=resid).to(dev())
    x = torch.randn(37, 49, device=dev())
    w = torch.randn(37, 10, device=dev())
    ops.PATH_LOG = set()
    try:
        mu, ls = enc(x)
        assert 'mlp_encoder.kernels' in ops.PATH_LOG
    finally:
        ops.PATH_LOG = None
    (torch.cat([mu, ls], 1) * w).sum().backward()
    got = {k_: t.grad.clone() for k_, t in enc.named_parameters()}
    enc.zero_grad()
    ref = enc.layers(x.double().float())
    assert rel_err(torch.cat([mu, ls], 1), ref) < OUT_TOL
    (ref * w).sum().backward()
    for k_, t in enc.named_parameters():
        assert rel_err(got[k_], t.grad) < GRAD_TOL, k_


@pytest.mark.parametrize('name,t_inf,r_inf', [('step_unimodal_unimodal', 'unimodal', 'unimodal'),
                                              ('step_attention_unimodal_gc4', 'attention', 'unimodal'),
                                              ('step_attention_unimodal_gc0', 'attention', 'unimodal')])
def test_secondary_branches_golden(name, t_inf, r_inf):
    """--t-inf/--r-inf unimodal branches of eval_minibatch (train_mnist.py:35-185) against the reference."""
    import src.models as M
    from tvae import step
    fx = load_golden(name)
    n, zd, gc = [int(v) for v in fx['cfg']]
    gen = M.SpatialGenerator(zd, 32, num_layers=2)
    if t_inf == 'unimodal':
        enc = M.InferenceNetwork_UnimodalTranslation_UnimodalRotation(n * n, zd + 3, 32, num_layers=2)
        noise = torch.from_numpy(fx['eps']).to(dev())
    else:
        enc = M.InferenceNetwork_AttentionTranslation_UnimodalRotation(n, 1, zd, kernels_num=8, groupconv=gc)
        noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    enc.load_state_dict(tdict(fx, 'e.'))
    gen.load_state_dict(tdict(fx, 'd.'))
    enc, gen = enc.to(dev()), gen.to(dev())
    from tvae import ops
    ops.PATH_LOG = set()
    try:
        elbo, logp, kl = step.eval_minibatch(O.image_coords(n).to(dev()), torch.from_numpy(fx['y']).to(dev()), gen, enc,
                                             t_inf, r_inf, 0, dev(), np.pi, gc, n, noise=noise)
        taken = ops.PATH_LOG
    finally:
        ops.PATH_LOG = None
    if t_inf == 'attention':                             # the whole encoder ran on the kernels
        assert ('trans_attn.rot_pool' if gc else 'trans_attn.plain') in taken, taken
    else:
        assert 'mlp_encoder.kernels' in taken, taken
    assert abs(float(elbo) - float(fx['elbo'])) / abs(float(fx['elbo'])) < OUT_TOL
    assert abs(float(logp) - float(fx['log_p'])) / abs(float(fx['log_p'])) < OUT_TOL
    assert abs(float(kl) - float(fx['kl'])) / abs(float(fx['kl'])) < OUT_TOL
    (-elbo).backward()
    assert_encoder_grads(enc, fx, GRAD_TOL)
    for k_, t in gen.named_parameters():
        assert_grad_close(t.grad, fx['gd.' + k_], tol=GRAD_TOL, name='gen.' + k_)


@pytest.mark.parametrize('name', sorted(HOT))
def test_inference_mode_forward_equals_training_forward(name, gemm_mode):
    """Under torch.no_grad() (eval_model, train_mnist.py:352-387; get_latent, clustering_mnist.py:121-161) the encoder and
    the decoder run their inference-mode forward (tvae/ops.py `_INFER`: no conv2 activation, no sign words, no decoder sign
    bits, nothing retained).  Same kernels minus stores: every output must be BITWISE the training-mode forward's, and the
    ELBO terms the reference's."""
    from tvae import ops, step
    fx = load_golden(name)
    lik, _ = HOT[name]
    enc, gen, n = seeded_models(fx)
    enc, gen = enc.to(dev()), gen.to(dev())
    x = O.image_coords(n).to(dev())
    y = torch.from_numpy(fx['y']).to(dev())
    noise = tuple(torch.from_numpy(fx[k_]).to(dev()) for k_ in ('E', 'eps_z', 'eps_theta'))
    et, lt, kt, aux_t = step.elbo_terms(x, y, gen, enc, lik, noise, return_aux=True)
    ops.PATH_LOG = set()
    try:
        with torch.no_grad():
            ei, li, ki, aux_i = step.elbo_terms(x, y, gen, enc, lik, noise, return_aux=True)
        took = set(ops.PATH_LOG)
    finally:
        ops.PATH_LOG = None
    assert 'enc.inference' in took, took
    assert not ({'dec.sign_bits', 'dec.no_h'} & took), took
    if gemm_mode in ('x6', 'h3') and name != 'hot_M50_B2':
        assert 'dec.no_h_inference' in took, took
    for a, b in ((et, ei), (lt, li), (kt, ki)):
        assert torch.equal(a.detach(), b)
    for k_ in ('heads', 'y_hat', 'z', 'theta', 'dx', 'a_sampled', 'q_t_r'):
        assert torch.equal(aux_t[k_].detach(), aux_i[k_]), k_
    assert abs(float(ei) - float(fx['elbo'])) / abs(float(fx['elbo'])) < OUT_TOL
    assert abs(float(li) - float(fx['log_p'])) / abs(float(fx['log_p'])) < OUT_TOL
    assert abs(float(ki) - float(fx['kl'])) / abs(float(fx['kl'])) < OUT_TOL
    # a later training step is unaffected by the switch (it is scoped to the no_grad calls)
    e2, _, _ = step.elbo_terms(x, y, gen, enc, lik, noise)
    assert torch.equal(e2.detach(), et.detach()) and e2.requires_grad


def test_decoder_pads_pixel_ranges_to_the_gemm_tile(gemm_mode):
    """Round 5: 28 x 28 images (784 pixels, not a multiple of the 128-column tile) reach the decoder's fast path -- recomputed
    first layer, unsaved last hidden activation, fused first-layer backward -- by padding every image's pixel range to 896
    (src/models.py: SpatialGenerator.forward, tvae/ops.py: decoder_padded_pixels).  Outputs and every gradient must agree with
    the unpadded path (different kernels: fp32-rounding level, not bitwise), and the padded branch must be the one taken."""
    import src.models as M
    from tvae import ops, step
    if gemm_mode == 'f32':
        pytest.skip('the fast path is a split-pipe path')
    torch.manual_seed(3)
    n, B = 28, 32
    gen = M.SpatialGenerator(2, 512, num_layers=2).to(dev())
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(n, 1, 2, kernels_num=128, kernels_size=28, padding=8,
                                                                    groupconv=8, rot_refinement=True, theta_prior=np.pi,
                                                                    normal_prior_over_r=False).to(dev())
    y = torch.rand(B, 1, n, n, device=dev())
    x = O.image_coords(n).to(dev())
    noise = step.draw_noise(B, 8 * 17 * 17, 2, dev())
    res = {}
    for pad in (True, False):
        saved = ops.DEC_PAD
        ops.DEC_PAD, ops.PATH_LOG = pad, set()
        try:
            for p in list(gen.parameters()) + list(enc.parameters()):
                p.grad = None
            elbo, lp, kl, aux = step.elbo_terms(x, y, gen, enc, 'bce', noise, return_aux=True)
            (-elbo).backward()
            torch.cuda.synchronize()
            res[pad] = (float(elbo), aux['y_hat'].detach().clone(), [p.grad.detach().clone() for p in gen.parameters()] +
                        [p.grad.detach().clone() for p in enc.parameters()], set(ops.PATH_LOG))
        finally:
            ops.DEC_PAD, ops.PATH_LOG = saved, None
    assert {'dec.virt_act', 'dec.no_h', 'dec.fuse_in'} <= res[True][3], res[True][3]
    assert 'dec.virt_act' not in res[False][3]
    assert tuple(res[True][1].shape) == (B, n * n, 1)
    assert abs(res[True][0] - res[False][0]) / abs(res[False][0]) < 1e-5
    assert rel_err(res[True][1], res[False][1]) < 1e-5
    names = [k_ for k_, _ in gen.named_parameters()] + [k_ for k_, _ in enc.named_parameters()]
    for nm, a, b in zip(names, res[True][2], res[False][2]):
        if nm == 'conv_a.bias':
            continue
        assert float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) < 2e-3, nm      # (kink flips between the two paths)


@pytest.mark.parametrize('fourier,n_out,resid', [(False, 1, False), (True, 3, False), (False, 3, True), (True, 1, False)])
def test_deep_decoder_runs_h3_on_measured_bounds(fourier, n_out, resid, gemm_mode):
    """Round 6: every hidden layer of a deep decoder (galaxy configuration: 4 layers, reference
    train_galaxy.py:412-420, src/models.py:84-93) runs the two-part h3 arithmetic -- the launch that stores a hidden activation
    (or a hidden gradient) leaves its measured maximum for the launch that streams it next (tvae_linear_fwd_x6 /
    tvae_linear_dgrad_x6 y_amax).  Against float64 on the CPU: outputs 1e-5, gradients 2e-4 of max-norm; the h3 instances
    must be the ones that ran (no silent x6), also for a hidden unit 2^-20 below its layer."""
    import src.models as M
    from tvae import ops
    if gemm_mode != 'h3':
        pytest.skip('h3 routing test')
    torch.manual_seed(11)
    zd, hid, L, B, Np = 4, 512, 4, 4, 256
    sigma = 2.0 / 15
    gen = M.SpatialGenerator(zd, hid, n_out=n_out, num_layers=L, resid=resid, fourier_expansion=fourier, sigma=sigma)
    with torch.no_grad():                                # a hidden unit far below the others (dead / untrained), layer 2
        lin = gen.layers[3].linear if resid else gen.layers[3]
        lin.weight[7].mul_(2.0 ** -20)
        lin.bias[7].mul_(2.0 ** -20)
    sd64 = {k_: v.detach().double().clone().requires_grad_(v.dtype.is_floating_point and 'embed' not in k_)
            for k_, v in gen.state_dict().items()}
    x = (torch.rand(B, Np, 2) * 2 - 1)
    z = torch.randn(B, zd)
    gy = torch.randn(B, Np, n_out)
    x64, z64 = x.double().requires_grad_(True), z.double().requires_grad_(True)
    import torch.nn.functional as F_nn

    def ref():                                           # oracle.generator_forward in float64 (same formulas, reference lines there)
        h = x64.reshape(B * Np, 2)
        if fourier:
            h = torch.cos(F_nn.linear(h, sd64['embed_latent.weight'] / sigma, sd64['embed_latent.bias']))
        h = F_nn.linear(h, sd64['coord_linear.weight'], sd64['coord_linear.bias']).view(B, Np, -1)
        h = (h + F_nn.linear(z64, sd64['latent_linear.weight']).unsqueeze(1)).view(B * Np, -1)
        h = F_nn.leaky_relu(h, 0.01)
        li = 1
        for _ in range(1, L):
            if resid:
                h = F_nn.leaky_relu(F_nn.linear(h, sd64[f'layers.{li}.linear.weight'], sd64[f'layers.{li}.linear.bias']) + h, 0.01)
                li += 1
            else:
                h = F_nn.leaky_relu(F_nn.linear(h, sd64[f'layers.{li}.weight'], sd64[f'layers.{li}.bias']), 0.01)
                li += 2
        return F_nn.linear(h, sd64[f'layers.{li}.weight'], sd64[f'layers.{li}.bias']).view(B, Np, n_out)

    y64 = ref()
    (y64 * gy.double()).sum().backward()
    gen = gen.to(dev())
    xg, zg = x.to(dev()).requires_grad_(True), z.to(dev()).requires_grad_(True)
    ops.PATH_LOG, ops.PARTS_LOG = set(), {}
    try:
        yh = gen(xg, zg)
        (yh * gy.to(dev())).sum().backward()
        torch.cuda.synchronize()
        took, plog = set(ops.PATH_LOG), dict(ops.PARTS_LOG)
    finally:
        ops.PATH_LOG, ops.PARTS_LOG = None, None
    assert {'dec.hidden_h3_meas', 'dec.dgrad_h3_meas', 'dec.wgrad_h3_meas'} <= took, took
    # every split-pipe launch of the decoder ran a two-part instance (the recomputed / two-valued forms included); the only
    # three-part launch left is a residual first hidden layer behind a stored coordinate layer (no h3 instance: x6)
    # (without Fourier features the first hidden layer's weight gradient meets a stored gradient with the RECOMPUTED coordinate
    #  layer: that operand pair has no h3 instance)
    for ep_ in ('tvae_linear_fwd_x6', 'tvae_linear_dgrad_x6', 'tvae_linear_wgrad_x6'):
        ps = [p_ for p_, _ in plog[ep_]]
        allowed3 = (1 if resid else 0) + (1 if (ep_ == 'tvae_linear_wgrad_x6' and not fourier and not resid) else 0)
        assert ps.count(3) <= allowed3 and set(ps) <= {2, 3}, (ep_, ps)
    assert rel_err(yh, y64.detach()) < 1e-5
    assert_grad_close(xg.grad.reshape(-1, 2), x64.grad.reshape(-1, 2), tol=2e-4, name='gx')
    assert_grad_close(zg.grad, z64.grad, tol=2e-4, name='gz')
    for k_, t in gen.named_parameters():
        assert_grad_close(t.grad, sd64[k_].grad, tol=2e-4, name=k_)
    # the dead unit's own gradient row, relative to ITSELF (its scale group is its row / its bound)
    wname = 'layers.3.linear.weight' if resid else 'layers.3.weight'
    g_dead, r_dead = dict(gen.named_parameters())[wname].grad[7].double().cpu(), sd64[wname].grad[7]
    assert float((g_dead - r_dead).abs().max() / r_dead.abs().max()) < 1e-3, 'dead unit row'


@pytest.mark.parametrize('fourier,L,n_out,resid', [(True, 2, 1, False), (False, 3, 1, False), (True, 4, 3, False), (False, 3, 3, True)])
def test_decoder_padded_row_stride_is_layout_only(fourier, L, n_out, resid, gemm_mode):
    """Round 6: decoders that store activations / gradients (Fourier first layer, two or more hidden layers; galaxy: reference
    train_galaxy.py:412-420, src/models.py:84-123) give their [features][B*Np] tensors a padded row stride when the column count
    is a multiple of 2^14 (ops._dec_ld: the rows of a weight-gradient panel otherwise share their L2 sets).  Layout only: output
    and every gradient bit for bit what the dense layout gives (TVAE_DEC_LD_PAD=0), and the padded path must be the one taken."""
    import src.models as M
    from tvae import ops
    torch.manual_seed(3)
    zd, hid, B, Np = 3, 512, 4, 4096                     # 2^14 columns
    gen = M.SpatialGenerator(zd, hid, n_out=n_out, num_layers=L, resid=resid, fourier_expansion=fourier, sigma=0.1).to(dev())
    x = (torch.rand(B, Np, 2, device=dev()) * 2 - 1)
    z = torch.randn(B, zd, device=dev())
    gy = torch.randn(B, Np, n_out, device=dev())

    def run(pad):
        old = ops.DEC_LD_PAD
        ops.DEC_LD_PAD = pad
        ops.PATH_LOG = set()
        try:
            for p_ in gen.parameters():
                p_.grad = None
            xg, zg = x.clone().requires_grad_(True), z.clone().requires_grad_(True)
            yh = gen(xg, zg)
            (yh * gy).sum().backward()
            torch.cuda.synchronize()
            return [yh.detach(), xg.grad, zg.grad] + [p_.grad.clone() for p_ in gen.parameters()], set(ops.PATH_LOG)
        finally:
            ops.DEC_LD_PAD, ops.PATH_LOG = old, None

    r1, took1 = run(64)
    r0, took0 = run(0)
    split = gemm_mode in ('x6', 'h3', 'bf16')
    assert ('dec.ld_pad' in took1) == split and 'dec.ld_pad' not in took0, (took1, took0)
    for a_, b_ in zip(r1, r0):
        assert torch.equal(a_, b_)
    with torch.no_grad():                                # inference path (no stored tensors besides the first layer's)
        ops.DEC_LD_PAD = 64
        try:
            yi = gen(x, z)
        finally:
            ops.DEC_LD_PAD = 64 if int(os.environ.get('TVAE_DEC_LD_PAD', '64')) else 0
    assert rel_err(yi, r0[0]) < 1e-6


@pytest.mark.parametrize('geom', [(20, 20, 4, 3), (21, 20, 3, 2)])
@pytest.mark.parametrize('zd', [10, 50])
def test_encoder_with_many_head_rows_takes_the_wide_tail(zd, geom, gemm_mode):
    """Round 6: z_dim > 2 gives 3 + 2 z_dim > 7 head rows (galaxy: 103, reference
    train_galaxy.py:412-420); conv2 + the stacked heads then run as two chained split-pipe GEMMs per direction
    (tvae_enc_tail_fwd_wide / _dgrad_wide) instead of five fp32-MFMA GEMMs.  The 7-tuple and every parameter gradient against
    the exact fp32-MFMA arithmetic of the same modules; the wide branch must be the one taken (h3), under no_grad too.
    Second geometry: a column count that is a multiple of 32 -- the two weight gradients then run as cooperative reductions
    (tvae_enc_tail_wgrad_wide) instead of fp32-MFMA GEMMs."""
    import src.models as M
    from tvae import ops
    from tvae._lib import arithmetic
    if gemm_mode != 'h3':
        pytest.skip('h3 routing test')
    torch.manual_seed(5)
    (n, k, pad, B), R = geom, 4
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(n, 1, zd, kernels_num=128, kernels_size=k, padding=pad,
                                                                    groupconv=R, rot_refinement=True, theta_prior=np.pi,
                                                                    normal_prior_over_r=False).to(dev())
    with torch.no_grad():
        for m in (enc.conv_a, enc.conv_r, enc.conv_z):
            m.weight.mul_(8.0)
    y = torch.rand(B, 1, n, n, device=dev())
    Ho = n + 2 * pad - k + 1
    gen = torch.Generator(device=dev()).manual_seed(1)
    wts = [torch.randn(B, R, Ho, Ho, device=dev(), generator=gen), torch.randn(B, 2, R, Ho, Ho, device=dev(), generator=gen),
           torch.randn(B, 2 * zd, R, Ho, Ho, device=dev(), generator=gen)]

    def run(mode):
        for p in enc.parameters():
            p.grad = None
        ops.PATH_LOG = set()
        try:
            with arithmetic(mode):
                torch.manual_seed(77)                    # the encoder draws its Gumbel noise itself (reference models.py:387)
                out = enc(y, dev())
                attn, q, p_r, a, off, theta, z = out
                ((q * wts[0]).sum() + (theta * wts[1]).sum() + (z * wts[2]).sum()).backward()
            torch.cuda.synchronize()
            return [t.detach().clone() for t in (attn, q, a, theta, z)], {k_: p.grad.clone() for k_, p in enc.named_parameters()}, set(ops.PATH_LOG)
        finally:
            ops.PATH_LOG = None

    o1, g1, took = run('h3')
    o0, g0, took0 = run('f32')
    assert {'enc.tail_fwd_wide', 'enc.tail_dgrad_wide'} <= took, took
    assert ('enc.tail_wgrad_wide' in took) == ((B * R * Ho * Ho) % 32 == 0), took
    assert 'enc.tail_fwd_wide' not in took0
    for a_, b_ in zip(o1, o0):
        assert rel_err(a_, b_) < 2e-5
    for k_ in g0:
        if k_ == 'conv_a.bias':
            continue                                     # analytically zero (the log-softmax is shift invariant): rounding noise
        assert_grad_close(g1[k_], g0[k_], tol=GRAD_TOL, name=k_)
    ops.PATH_LOG = set()
    try:
        with torch.no_grad():
            torch.manual_seed(77)
            oi = enc(y, dev())
        took_i = set(ops.PATH_LOG)
    finally:
        ops.PATH_LOG = None
    assert {'enc.tail_fwd_wide', 'enc.inference'} <= took_i, took_i
    assert torch.equal(oi[5], o1[3]) and torch.equal(oi[6], o1[4])
