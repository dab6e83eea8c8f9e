"""Conditioning of the galaxy-shape step (BASELINE configs[4], B = 2): how far do the gradients move
(a) between the default arithmetic (frequency-domain convolution, exact-split products) and exact fp32 products, and
(b) inside ONE arithmetic when the input is perturbed by 1e-6 relative (LeakyReLU kink flips among 2 x 68 M encoder
pre-activations)?  If (b) is as large as (a), (a) is conditioning, not arithmetic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import numpy as np, torch
import src.models as M
from tvae import _lib, ops, step, tables
dev = torch.device('cuda:0')
torch.manual_seed(0)
n, R, B, zd = 128, 16, 2, 50
gen = M.SpatialGenerator(zd, 512, n_out=3, num_layers=4, fourier_expansion=True, sigma=2.0 / (n - 1)).to(dev)
enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(n, 3, zd, kernels_num=128, kernels_size=64, padding=32,
    groupconv=R, rot_refinement=True, theta_prior=np.pi, normal_prior_over_r=False).to(dev)
Ho = n + 2 * 32 - 64 + 1
y = torch.rand(B, 3, n, n, device=dev)
x = torch.from_numpy(tables.image_coords(n)).to(dev)
noise = step.draw_noise(B, R * Ho * Ho, zd, dev)
params = list(enc.named_parameters()) + list(gen.named_parameters())
def run(mode, yy):
    for _, p in params:
        p.grad = None
    with _lib.arithmetic(mode):
        e, lp, kl = step.elbo_terms(x, yy, gen, enc, 'bce3', noise)
    (-e).backward()
    return float(e), {nm: p.grad.clone() for nm, p in params}
e_f, g_f = run('f32', y)
e_x, g_x = run('x6', y)
e_p, g_p = run('f32', y * (1 + 1e-6 * torch.randn_like(y)))
e_q, g_q = run('x6', y * (1 + 1e-6 * torch.randn_like(y)))
print('elbo f32 %.6f  x6 %.6f  f32-perturbed %.6f  x6-perturbed %.6f' % (e_f, e_x, e_p, e_q))
for nm in g_f:
    s = float(g_f[nm].abs().max()) + 1e-30
    print('%-28s x6-vs-f32 %.2e   f32 perturbed %.2e   x6 perturbed %.2e' % (
        nm, float((g_x[nm] - g_f[nm]).abs().max()) / s, float((g_p[nm] - g_f[nm]).abs().max()) / s,
        float((g_q[nm] - g_x[nm]).abs().max()) / s))
# forward agreement of the convolution alone
with torch.no_grad():
    outs = {}
    for mode in ('f32', 'x6'):
        with _lib.arithmetic(mode):
            outs[mode] = ops.conv1_forward(y, enc.conv1.weight, enc.conv1.bias, 128, R, 64, 32, 1)
    d = (outs['x6'] - outs['f32']).double()
    print('conv1 output: relative Frobenius difference x6(DFT) vs f32 %.3e, max abs %.3e (max |out| %.3e)' % (
        float(d.norm() / outs['f32'].double().norm()), float(d.abs().max()), float(outs['f32'].abs().max())))
