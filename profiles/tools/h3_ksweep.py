#!/usr/bin/env python3
"""time(K) = a + b K of the decoder's forward launch (recomputed first-layer operand, 512 rows x 256*4096 columns) in the x6
and h3 arithmetics: the slope is the cost of a 16-k step, the intercept what a tile costs outside its k-loop (prologue,
epilogue, launch) -- and of its data-gradient (two-valued) launch.   python profiles/tools/h3_ksweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import torch
from tvae._lib import call, query

dev = torch.device('cuda', 0)
B, Np, M = 256, 4096, 512
Nt = B * Np
g = torch.Generator().manual_seed(0)
xr = torch.randn(Nt, 2, generator=g).to(dev)
Y = torch.empty(M, Nt, device=dev)
bits = torch.empty(M, Nt // 32, dtype=torch.int32, device=dev)
yh = torch.empty(Nt, device=dev)
for K in (128, 256, 384, 512):
    Wc, bc = torch.randn(K, 2, generator=g).to(dev), torch.randn(K, generator=g).to(dev)
    LB = torch.randn(B, K, generator=g).to(dev)
    W, b = (torch.randn(M, K, generator=g) * K ** -0.5).to(dev), torch.randn(M, generator=g).to(dev)
    wo, bo = torch.randn(M, generator=g).to(dev), torch.zeros(1, device=dev)
    for nparts, split in ((3, 'tvae_dense_split3'), (2, 'tvae_dense_split2h')):
        w3 = torch.empty(query('tvae_dense_x6_bytes', M, K) // 4, device=dev)
        call(split, W, K, w3, w3.numel() * 4, M, K, 0, None, None)
        for tag, extra in (('plain', (None, None, None)), ('coldot+bits', (wo, bo, yh))):
            def run():
                call('tvae_linear_fwd_x6', w3, None, b, None, Y, M, Nt, K, Nt, Nt, 1, 0.01, *extra, xr, Wc, bc, LB, Np,
                     bits if tag != 'plain' else None, nparts)
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            print(f'K {K:4d} parts {nparts} {tag:12s} {e0.elapsed_time(e1) / 5:7.3f} ms', flush=True)
