"""Debug: deep decoder gradients under TVAE_H3_DEEP on / off against the fp32-MFMA arithmetic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import torch
import src.models as M
from tvae import ops
from tvae._lib import arithmetic
dev = torch.device('cuda:0')
torch.manual_seed(1)
zd, hid, L, B, Np, n_out = 50, 512, 4, 2, 1024, 3
fourier = os.environ.get('FOURIER', '1') == '1'
gen = M.SpatialGenerator(zd, hid, n_out=n_out, num_layers=L, fourier_expansion=fourier, sigma=2.0 / 127).to(dev)
x = (torch.rand(B, Np, 2, device=dev) * 2 - 1)
z = torch.randn(B, zd, device=dev)
gy = torch.randn(B, Np, n_out, device=dev) * 1e-3


def run(mode, deep):
    ops.H3_DEEP = deep
    for p in gen.parameters():
        p.grad = None
    xg, zg = x.clone().requires_grad_(True), z.clone().requires_grad_(True)
    ops.PATH_LOG = set()
    with arithmetic(mode):
        yh = gen(xg, zg)
        (yh * gy).sum().backward()
    torch.cuda.synchronize()
    took = sorted(ops.PATH_LOG); ops.PATH_LOG = None
    out = {'y': yh.detach().clone(), 'gx': xg.grad.clone(), 'gz': zg.grad.clone()}
    out.update({k: p.grad.clone() for k, p in gen.named_parameters()})
    return out, took


ref, _ = run('f32', False)
for deep in (False, True):
    got, took = run('h3', deep)
    print('H3_DEEP', deep, took)
    for k in ref:
        e = float((got[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-30))
        print('   %-28s rel %.3e %s' % (k, e, '  <<<<' if e > 1e-3 else ''))
