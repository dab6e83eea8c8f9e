"""Fixed per-tile cost of the split-pipe dense GEMM: time(K) = a + b*K at M = 512, N = 2^20 (8192 column tiles).
The intercept a is what a tile pays outside its k-loop (prologue loads, epilogue stores, launch tail)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import torch
import tvae._lib as _L
if os.environ.get('TVAE_LIB'):
    _L.LIB_PATH = os.path.abspath(os.environ['TVAE_LIB'])
from tvae._lib import call, query
dev = torch.device('cuda:0')
M, Nt = 512, 1 << 20
reps = 5
bb = torch.randn(M, device=dev)
out = torch.empty(M, Nt, device=dev)
res = []
for K in [int(v) for v in os.environ.get('KS', '128,256,512,1024,2048').split(',')]:
    W = torch.randn(M, K, device=dev) * 0.05
    X = torch.randn(K, Nt, device=dev)
    w3 = torch.empty(query('tvae_dense_x6_bytes', M, K) // 4, device=dev)
    call('tvae_dense_split3', W, K, w3, w3.numel() * 4, M, K, 0, None, None)
    for name, act, store in (('fwd_lrelu', 1, True),):
        fn = lambda: call('tvae_linear_fwd_x6', w3, X, bb, None, out, M, Nt, K, Nt, Nt, act, 0.01, None, None, None, None, None, None, None, 0, None, 3)
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        fl = 2.0 * M * K * Nt
        print(f'{name} K={K:5d} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s alg  ({6 * fl / ms / 1e12:6.3f} PF/s bf16)', flush=True)
        res.append((K, ms))
    del X, W
(k0, t0), (k1, t1) = res[-2], res[-1]
b = (t1 - t0) / (k1 - k0)
print('slope %.4f ms per 512 k  -> %.1f TF/s alg in the k-loop;  intercepts:' % (b * 512, 2.0 * M * 512 * Nt / (b * 512) / 1e9),
      ' '.join('K=%d: %.3f ms' % (k, t - b * k) for k, t in res))
