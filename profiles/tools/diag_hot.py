"""Diagnostic: per-tensor gradient errors of the hot-width fixtures in every arithmetic mode, against the reference
golden and against the oracle (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'target-vae_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
from conftest import load_golden, seeded_models
from oracle import tvae_oracle as O
from tvae import _lib, step

dev = torch.device('cuda:0')
for name, lik in (('hot_S28F_B8', 'bce'), ('hot_S64_B2', 'gauss')):
    fx = load_golden(name)
    cfgv = [int(v) for v in fx['cfg']]
    enc0, gen0, n = seeded_models(fx)
    # oracle with aux + grads
    encp = {k: v.detach().clone().requires_grad_(True) for k, v in enc0.state_dict().items()}
    genp = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in gen0.state_dict().items()}
    nz = dict(E=torch.from_numpy(fx['E']), eps_z=torch.from_numpy(fx['eps_z']), eps_theta=torch.from_numpy(fx['eps_theta']))
    e, lp, kl, aux = O.elbo_step(O.image_coords(n), torch.from_numpy(fx['y']), encp, genp, R=cfgv[6], padding=cfgv[5],
                                 rot_refinement=True, theta_prior=float(fx['theta_prior']), normal_prior_over_r=False,
                                 num_layers=cfgv[10], resid=False, fourier_sigma=float(fx['sigma']) if cfgv[12] else None,
                                 likelihood=lik, return_aux=True, **nz)
    for k in ('z', 'theta', 'dx', 'y_hat'):
        aux[k].retain_grad() if aux[k].requires_grad else None
    (-e).backward()
    for mode in ('f32', 'x6'):
        _lib.set_gemm_mode(mode)
        enc, gen, _ = seeded_models(fx)
        enc, gen = enc.to(dev), gen.to(dev)
        x = O.image_coords(n).to(dev)
        noise = tuple(t.to(dev) for t in nz.values())
        elbo, logp, klg, got = step.elbo_terms(x, torch.from_numpy(fx['y']).to(dev), gen, enc, lik, noise, return_aux=True)
        for k in ('z', 'theta', 'dx', 'y_hat', 'heads'):
            if got[k].requires_grad:
                got[k].retain_grad()
        (-elbo).backward()
        print(f'== {name} mode {mode}: elbo {float(elbo):.6f} ref {float(fx["elbo"]):.6f} oracle {float(e):.6f}')
        for k in ('z', 'theta', 'dx', 'y_hat', 'a_sampled', 'q_t_r', 'kl_per_image'):
            a, b = got[k].detach().cpu().double().reshape(-1), aux[k].detach().double().reshape(-1)
            print(f'   fwd {k:14s} rel {float((a-b).abs().max()/b.abs().max()):.3e}')
        for k in ('z', 'theta', 'dx', 'y_hat'):
            if got[k].grad is not None and aux[k].grad is not None:
                a, b = got[k].grad.cpu().double().reshape(-1), aux[k].grad.double().reshape(-1)
                print(f'   grad wrt {k:10s} rel {float((a-b).abs().max()/b.abs().max()):.3e}  max|b| {float(b.abs().max()):.3e}')
        for pre, mod, orc in (('ge.', enc, encp), ('gd.', gen, genp)):
            for k, t in mod.named_parameters():
                a = t.grad.cpu().double()
                b = torch.from_numpy(fx[pre + k]).double()
                c = orc[k].grad.double()
                sc = float(b.abs().max())
                print(f'   {pre}{k:24s} gpu-ref {float((a-b).abs().max())/sc:.3e}  gpu-oracle {float((a-c).abs().max())/sc:.3e}'
                      f'  oracle-ref {float((c-b).abs().max())/sc:.3e}  max|b| {sc:.3e}')
