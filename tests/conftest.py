import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'target-vae_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible; selection is by -m."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: d[k] for k in d.files}


def tdict(fx, prefix, requires_grad=False):
    out = {}
    for k, v in fx.items():
        if k.startswith(prefix):
            t = torch.from_numpy(np.array(v))
            if requires_grad and t.dtype.is_floating_point:
                t.requires_grad_(True)
            out[k[len(prefix):]] = t
    return out


def rel_err(a, b):
    """max |a-b| / max|b| (tensor-level relative error, SURVEY 8d parity gate)."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    den = b.abs().max().item()
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)


def assert_grad_close(a, b, tol=2e-4, floor=0.0, bad_rows=0.02, bad_tol=0.3, name=''):
    """Gradient comparison that tolerates rare LeakyReLU-kink flips.

    A pre-activation within rounding of 0 can land on either side of the kink in two
    implementations; one flipped element perturbs one output-channel row of the upstream
    weight gradient by a few percent.  Rows (dim 0) are compared in max-norm relative to
    max|b| (or `floor` for analytically-zero grads such as conv_a.bias, whose softmax is
    shift-invariant); at most max(1, bad_rows*rows) rows may exceed `tol`, none `bad_tol`.
    The allowance only exists for tensors with several output-channel rows: a single-row tensor
    (conv_a.weight (1,C,..), the decoder's Wo (1,hid), any bias vector seen as one row) must meet
    `tol` outright -- its one row IS the whole gradient.
    """
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(b.abs().max().item(), floor, 1e-30)
    err = ((a - b).abs() / scale).reshape(a.shape[0] if a.dim() > 0 else 1, -1).max(dim=1).values
    nbad = int((err > tol).sum())
    allowed = 0 if err.numel() < 4 else max(1, int(np.ceil(bad_rows * err.numel())))
    assert nbad <= allowed and err.max().item() < bad_tol, \
        f'{name}: {nbad}/{err.numel()} rows > {tol} (allowed {allowed}), max {err.max().item():.3e}'


def assert_encoder_grads(enc, fx, tol, prefix='ge.', kink_prefix=None, name='enc.', elbo_loss=True):
    """Every encoder parameter gradient against the fixture.  `floor` = 1e-3 of the largest encoder gradient entry (tensors
    whose gradients are tiny next to the others are compared on that scale).  `conv_a.bias` has an analytically ZERO
    gradient (the softmax over its logits is shift invariant): both sides hold only the rounding noise of a cancelling
    sum, so they are not compared with each other -- each must stay below 1e-5 of the largest gradient entry.
    That holds for the ELBO (elbo_loss=True); a probe loss that reads `attn` directly gives conv_a.bias a real gradient.
    kink_prefix: per-tensor conditioning the fixture measured on the reference itself (make_goldens.py:gen_hotpath)."""
    gmax = max(float(np.abs(v).max()) for k_, v in fx.items() if k_.startswith(prefix))
    for k_, t in enc.named_parameters():
        want = fx[prefix + k_]
        if k_ == 'conv_a.bias' and elbo_loss:
            got = float(t.grad.detach().abs().max())
            assert got <= 1e-5 * gmax and float(np.abs(want).max()) <= 1e-5 * gmax, (name + k_, got, want, gmax)
            continue
        tol_k = max(tol, 2 * float(fx[kink_prefix + k_])) if kink_prefix else tol
        assert_grad_close(t.grad, want, tol=tol_k, floor=1e-3 * gmax, name=name + k_)


def assert_trajectory_update_close(enc, gen, fx, tol, cos_min=0.999):
    """Final state of a multi-step run (dicts / state_dicts `enc`, `gen`) against fixture entries eT.* / dT.* with the
    initial state e.* / d.*: relative L2 error and cosine of the total update p_T - p_0, per tensor.  conv_a.bias has an
    analytically zero gradient (softmax shift invariance): Adam follows rounding noise at up to lr per step, so it is only
    bounded by lr * T."""
    T = fx['curve'].shape[0]
    for pre, d in (('e', enc), ('d', gen)):
        for k_, v in d.items():
            if (pre + 'T.' + k_) not in fx:
                continue                                   # buffers
            w = torch.from_numpy(fx[pre + 'T.' + k_]).double()
            w0 = torch.from_numpy(fx[pre + '.' + k_]).double()
            got = torch.as_tensor(v).detach().double().cpu()
            if pre + '.' + k_ == 'e.conv_a.bias':
                assert (got - w).abs().max() <= 2 * float(fx['lr']) * T + 1e-7
                continue
            du, dr = (got - w0).reshape(-1), (w - w0).reshape(-1)
            rel = float((du - dr).norm() / dr.norm())
            cos = float(torch.dot(du, dr) / (du.norm() * dr.norm()))
            assert rel < tol and cos > cos_min, (pre + '.' + k_, rel, cos)


def seeded_models(fx):
    """Models of a seed-based fixture (tests/golden/make_goldens.py:gen_hotpath): default init of the drop-in
    src.models classes under fx['seed'] (generator first, then encoder, head weights scaled), verified entry by entry
    against the digests the reference-side construction stored.  Returns (encoder, generator, n) on the CPU."""
    import src.models as M
    n, cin, zd, C, k, p, R, refine, normal, hid, L, n_out, fourier, resid = [int(v) for v in fx['cfg']]
    torch.manual_seed(int(fx['seed']))
    gen = M.SpatialGenerator(zd, hid, n_out=n_out, num_layers=L, resid=bool(resid), fourier_expansion=bool(fourier),
                             sigma=float(fx['sigma']))
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, cin, zd, kernels_num=C, kernels_size=k, padding=p, groupconv=R, rot_refinement=bool(refine),
        theta_prior=float(fx['theta_prior']), normal_prior_over_r=bool(normal))
    with torch.no_grad():
        for nm in ('conv_a', 'conv_r', 'conv_z'):
            getattr(enc, nm).weight.mul_(float(fx['scale_heads']))
    for prefix, mod in (('se.', enc), ('sd.', gen)):
        sd = mod.state_dict()
        assert sorted(sd) == sorted(k_[len(prefix):] for k_ in fx if k_.startswith(prefix))
        for k_, v in sd.items():
            t = v.detach().double().reshape(-1)
            got = np.array([float(t.sum()), float(t.abs().sum()), float(t[0]), float(t[-1])])
            want = fx[prefix + k_]
            # the float64 sums depend on the reduction order (thread count) in the last bits; first / last are exact
            assert np.array_equal(got[2:], want[2:]) and np.allclose(got[:2], want[:2], rtol=0, atol=1e-11 * want[1]), \
                (k_, got, want)
    return enc, gen, n
