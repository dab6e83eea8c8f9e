"""Probe (round 6): do the encoder-tail kernels care about the row stride of their [128][N] tensors?  N = 256 x 8 x 33 x 33 columns
(row stride 2^13 x 1089 bytes) against the same problem with 64 floats of row padding."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
from tvae._lib import call, query
from tvae.ops import _enc_tail_perm
dev = torch.device('cuda:0')
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
C, nh, N = 128, 7, 256 * 8 * 33 * 33
W2 = torch.randn(C, C, device=dev) * C ** -0.5; b2 = torch.randn(C, device=dev)
Wh = torch.randn(nh, C, device=dev) * C ** -0.5; bh = torch.randn(nh, device=dev)
w3 = torch.empty(query('tvae_dense_x6_bytes', C, C) // 4, device=dev)
call('tvae_dense_split2h', W2, C, w3, w3.numel() * 4, C, C, 0, None, None)
w3p = torch.empty(query('tvae_dense_x6_bytes', C, C) // 4, device=dev)
call('tvae_dense_split2h', W2.t()[:, _enc_tail_perm(dev)].contiguous(), C, w3p, w3p.numel() * 4, C, C, 0, None, None)
wh3 = torch.empty(query('tvae_dense_x6_bytes', C, nh) // 4, device=dev)
call('tvae_dense_split3', Wh, C, wh3, wh3.numel() * 4, C, nh, 1, None, None)
for pad in (0, 64, 96):
    ld = N + pad
    A1 = torch.randn(C, ld, device=dev); H = torch.empty(C, ld, device=dev); heads = torch.empty(nh, ld, device=dev)
    dA1 = torch.empty(C, ld, device=dev); dheads = torch.randn(nh, ld, device=dev)
    bits = torch.zeros(2, N, 4, dtype=torch.int32, device=dev)
    a1max = A1.abs().amax(dim=1).contiguous()
    f = t(lambda: call('tvae_enc_tail_fwd_x6', w3, A1, ld, b2, Wh, bh, nh, H, ld, heads, ld, bits[0], bits[1], C, N, 1, 0.01, 2, a1max))
    d = t(lambda: call('tvae_enc_tail_dgrad_x6', w3p, wh3, dheads, ld, nh, bits[0], bits[1], dA1, ld, C, N, 0.01, 2))
    ws = torch.empty(query('tvae_enc_tail_wgrad_x6_ws_floats', N), device=dev); dW2 = torch.empty(C, C, device=dev)
    w = t(lambda: call('tvae_enc_tail_wgrad_x6', A1, ld, dheads, ld, nh, bits[0], Wh, dW2, ws, ws.numel(), C, N, 0.01, 2, a1max))
    print('ld = N + %3d: enc_tail fwd %.3f ms  dgrad %.3f ms  wgrad %.3f ms' % (pad, f, d, w))
