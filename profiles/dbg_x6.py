import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import torch
from tvae._lib import call, query
dev = torch.device('cuda:0')
R, Cin, n, k, pad = 8, 1, 64, 64, 16
Ho = 33
for (B, C) in [(17, 48), (32, 128), (64, 128), (256, 32), (256, 128)]:
    torch.manual_seed(0)
    y = torch.randn(B, Cin, n, n, device=dev)
    bank = torch.randn(C * R, k * k, device=dev) * 0.02
    bias = torch.randn(C, device=dev)
    N = B * R * Ho * Ho
    a3 = torch.empty(query('tvae_conv1_x6_bank_bytes', C, R, Cin, k) // 4, device=dev)
    call('tvae_bank_split3', bank, a3, a3.numel() * 4, C, R, Cin, k)
    o1 = torch.empty(C, N, device=dev); o2 = torch.empty(C, N, device=dev)
    call('tvae_conv1_fwd', y, bank, bias, o1, B, Cin, n, k, pad, C, R, 1, 0.01)
    call('tvae_conv1_fwd_x6', y, a3, bias, o2, B, Cin, n, k, pad, C, R, 1, 0.01)
    d = (o1 - o2).view(C, B, R, Ho * Ho)
    print(B, C, 'rel', float(d.norm() / o1.norm()), 'bad c', (d.abs().amax(dim=(1, 2, 3)) > 1e-3).nonzero().flatten().tolist()[:10],
          'bad b', (d.abs().amax(dim=(0, 2, 3)) > 1e-3).nonzero().flatten().tolist()[:10], 'bad r', (d.abs().amax(dim=(0, 1, 3)) > 1e-3).nonzero().flatten().tolist())
