import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import torch
dev = torch.device('cuda:0')
dbg = torch.zeros(64 * 8, dtype=torch.int64, device=dev)
os.environ['TVAE_DBGPTR'] = str(dbg.data_ptr())
import tvae._lib as _L
_L.LIB_PATH = os.path.join(ROOT, 'target-vae_amd/csrc/build/dbg.so')
from tvae._lib import call, query
F_, Nt = 512, 256 * 4096
h1 = torch.randn(F_, Nt, device=dev); h2 = torch.empty(F_, Nt, device=dev)
W = torch.randn(F_, F_, device=dev) * 0.05; bb = torch.randn(F_, device=dev)
w3 = torch.empty(query('tvae_dense_x6_bytes', F_, F_) // 4, device=dev)
call('tvae_dense_split3', W, F_, w3, w3.numel() * 4, F_, F_, 0)
for _ in range(2):
    call('tvae_linear_fwd_x6', w3, h1, bb, None, h2, F_, Nt, F_, Nt, Nt, 1, 0.01)
torch.cuda.synchronize()
d = dbg.view(64, 8).cpu()
print('block: loop cycles, nk, per-step, epilogue')
for bk in range(8):
    r = d[bk]
    print(bk * 2000, int(r[1] - r[0]), int(r[3]), int(r[1] - r[0]) // max(int(r[3]), 1), int(r[2] - r[1]))
