#!/usr/bin/env python3
"""x6 (three bf16 parts, six products) vs h3 (two fp16 parts, three products) on the frequency-domain lifting convolution at
the bench shape: time per call and relative error against an fp64 convolution of a batch slice.
  python profiles/tools/h3_conv_ab.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import torch
import torch.nn.functional as F
from tvae._lib import call, query

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, k, pad, C, R, Cin = 64, 64, 16, 128, 8, 1
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(1)
y = torch.randn(B, Cin, n, n, generator=g).to(dev)
bank = (torch.randn(C * R, Cin * k * k, generator=g) * (k * k) ** -0.5).to(dev)
bias = (torch.randn(C, generator=g) * 0.1).to(dev)
Ho = n + 2 * pad - k + 1
at = torch.zeros(query('tvae_conv1_dft_at_floats', B, Cin, n, k, pad, C, R), device=dev)
ws = torch.empty(query('tvae_conv1_dft_ws_floats', B, Cin, n, k, pad, C, R), device=dev)
out = torch.empty(C, B * R * Ho * Ho, device=dev)
dpre = torch.randn(C, B * R * Ho * Ho, generator=g).to(dev) * 1e-3
dbank = torch.empty(C * R, Cin * k * k, device=dev)
dbias = torch.empty(C, device=dev)
nb = 4
ref = F.conv2d(y[:nb].double(), bank.double().view(C * R, Cin, k, k), None, 1, pad).view(nb, C, R, Ho, Ho) + bias.double().view(1, C, 1, 1, 1)
ref = torch.where(ref > 0, ref, 0.01 * ref)
gsl = dpre.view(C, B, R, Ho, Ho)
ref_g = None
res = {}
for parts in (3, 2, 3, 2):
    for fn in ('fwd', 'wgrad'):
        def run():
            if fn == 'fwd':
                call('tvae_conv1_fwd_dft', y, bank, bias, out, at, ws, ws.numel(), B, Cin, n, k, pad, C, R, 1, 0.01, parts)
            else:
                call('tvae_conv1_wgrad_dft', dpre, at, dbank, dbias, ws, ws.numel(), B, Cin, n, k, pad, C, R, parts)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        if fn == 'fwd':
            got = out.view(C, B, R, Ho, Ho)[:, :nb].permute(1, 0, 2, 3, 4).double()
            err = float((got - ref).norm() / ref.norm())
            res[(parts, fn)] = out.clone()
        else:
            res[(parts, fn)] = dbank.clone()
            err = float('nan')
        print(f'parts {parts} {fn:5s} {ms:7.3f} ms   rel err vs fp64 {err:.3e}', flush=True)
d = float((res[(2, 'fwd')].double() - res[(3, 'fwd')].double()).norm() / res[(3, 'fwd')].double().norm())
print('forward h3 vs x6:', f'{d:.3e}')
d = float((res[(2, 'wgrad')].double() - res[(3, 'wgrad')].double()).norm() / res[(3, 'wgrad')].double().norm())
print('wgrad   h3 vs x6:', f'{d:.3e}')
