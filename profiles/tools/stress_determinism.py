#!/usr/bin/env python3
"""Bitwise-repeatability stress test of one training step under GPU sharing.

Every kernel of the library uses fixed reduction orders (no float atomics), so forward + backward on fixed inputs must give
BITWISE identical gradients every time.  This script runs the step `--iters` times in each of `--procs` concurrently
running processes (several processes time-slice ONE GPU: the condition under which tests/test_dp_gpu.py showed a rare
deviation) and reports every tensor that ever differs from the first iteration's value, with the iteration and the size
of the deviation.  Usage (GPU box):  python profiles/tools/stress_determinism.py --procs 2 --iters 300 [--cfg dp|S28|S64]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import numpy as np
import torch
import torch.multiprocessing as mp

CFGS = {   # n, zd, R, B, C, hidden, k, pad
    'dp': (28, 2, 8, 3, 16, 64, 28, 8),          # the shapes of tests/test_dp_gpu.py (fp32-MFMA GEMM paths)
    'S28': (28, 2, 8, 16, 128, 512, 28, 8),      # full widths (split-pipe paths)
    'S64': (64, 2, 8, 4, 128, 512, 64, 16),
}


def worker(rank, cfg, iters, out):
    import src.models as M
    from tvae import ops, step, tables
    n, zd, R, B, C, hid, k, pad = CFGS[cfg]
    dev = torch.device('cuda', 0)
    torch.manual_seed(3)
    gen = M.SpatialGenerator(zd, hid, num_layers=2).to(dev)
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, 1, zd, kernels_num=C, kernels_size=k, padding=pad, groupconv=R, rot_refinement=True, theta_prior=np.pi,
        normal_prior_over_r=False).to(dev)
    with torch.no_grad():
        for m in (enc.conv_a, enc.conv_r, enc.conv_z):
            m.weight.mul_(10.0)
    g = torch.Generator().manual_seed(11)
    y = torch.rand(B, 1, n, n, generator=g).to(dev)
    ho = n + 2 * pad - k + 1
    noise = (torch.empty(B, R * ho * ho).exponential_(generator=g).to(dev), torch.randn(B, zd, generator=g).to(dev),
             torch.randn(B, generator=g).to(dev))
    x = torch.from_numpy(tables.image_coords(n)).to(dev)
    params = [('d.' + k_, p) for k_, p in gen.named_parameters()] + [('e.' + k_, p) for k_, p in enc.named_parameters()]
    ref = None
    bad = {}
    for it in range(iters):
        for _, p in params:
            p.grad = None
        e, lp, kl, aux = step.elbo_terms(x, y, gen, enc, 'bce' if cfg != 'S64' else 'gauss', noise, return_aux=True)
        keep = {k_: v for k_, v in aux.items() if torch.is_tensor(v) and v.requires_grad}
        for v in keep.values():
            v.retain_grad()
        (-e).backward()
        cur = {nm: p.grad.clone() for nm, p in params}
        cur['elbo'] = e.detach().clone().reshape(1)
        for k_, v in aux.items():                        # intermediates of the forward, and the gradients that reach them
            if torch.is_tensor(v):
                cur['fwd.' + k_] = v.detach().clone().float()
        for k_, v in keep.items():
            if v.grad is not None:
                cur['bwd.' + k_] = v.grad.detach().clone().float()
        if ref is None:
            torch.cuda.synchronize()
            ref = cur
            continue
        for nm in cur:
            if not torch.equal(cur[nm], ref[nm]):
                d = float((cur[nm].double() - ref[nm].double()).abs().max() / ref[nm].double().abs().max().clamp_min(1e-30))
                nbad = int((cur[nm] != ref[nm]).sum())
                bad.setdefault(nm, []).append((it, d, nbad, cur[nm].numel()))
    torch.cuda.synchronize()
    with open(out + f'.{rank}', 'w') as f:
        for nm, ev in bad.items():
            f.write(f'rank {rank} {nm}: {len(ev)} deviating iterations, first {ev[:4]}\n')
        f.write(f'rank {rank} done: {iters} iterations, {len(bad)} tensors ever deviated\n')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--procs', type=int, default=2)
    ap.add_argument('--iters', type=int, default=300)
    ap.add_argument('--cfg', default='dp')
    ap.add_argument('--out', default='/tmp/stress_det')
    a = ap.parse_args()
    mp.start_processes(worker, args=(a.cfg, a.iters, a.out), nprocs=a.procs, join=True, start_method='spawn')
    for r in range(a.procs):
        print(open(a.out + f'.{r}').read(), end='')
