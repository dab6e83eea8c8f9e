"""Per-phase shader-clock totals of dense16_kernel<0> from the diagnostic build (libtvae_stamps.so, -DTVAE_D16_STAMPS):
prologue / k-loop / epilogue cycles per tile, median over workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import numpy as np
import torch
import tvae._lib as _L
_L.LIB_PATH = os.path.join(ROOT, 'target-vae_amd', 'csrc', 'build', os.environ.get('STAMPLIB', 'libtvae_stamps.so'))
from tvae._lib import call, query, lib
dev = torch.device('cuda:0')
M, Nt = 512, 1 << int(os.environ.get('LOGN', '20'))
L = lib()
buf = (ctypes.c_ulonglong * (256 * 4))()
for K in [int(v) for v in os.environ.get('KS', '512,2048').split(',')]:
    W = torch.randn(M, K, device=dev) * 0.05
    X = torch.randn(K, Nt, device=dev)
    bb = torch.randn(M, device=dev)
    out = torch.empty(M, Nt, device=dev)
    w3 = torch.empty(query('tvae_dense_x6_bytes', M, K) // 4, device=dev)
    call('tvae_dense_split3', W, K, w3, w3.numel() * 4, M, K, 0)
    fn = lambda: call('tvae_linear_fwd_x6', w3, X, bb, None, out, M, Nt, K, Nt, Nt, 1, 0.01, None, None, None, None, None, None, None, 0)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.tvae_debug_d16_stamps(None, 1)
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    L.tvae_debug_d16_stamps(buf, 0)
    a = np.array(buf[:], dtype=np.float64).reshape(256, 4)
    tiles = a[:, 3]
    per = a[:, :3] / np.maximum(tiles[:, None], 1)
    med = np.median(per, axis=0)
    print(f'K={K}: {s.elapsed_time(e):.3f} ms; tiles/WG {tiles.min():.0f}-{tiles.max():.0f}; cycles per tile (median over WGs): '
          f'prologue {med[0]:.0f}  k-loop {med[1]:.0f} ({med[1] / (K / 32):.0f} per 32-k step)  epilogue+drain {med[2]:.0f};  '
          f'sum x tiles = {np.median(a[:, :3].sum(1)) / 1e6:.2f} Mcycles (100 MHz ticks? see clock)')
