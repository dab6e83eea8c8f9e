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
B, C, R, Cin, n, k, pad = 256, 128, 8, 1, 64, 64, 16
Ho = 33
y = torch.randn(B, Cin, n, n, device=dev)
bank = torch.randn(C * R, k * k, device=dev) * 0.02
bias = torch.randn(C, device=dev)
a3 = torch.empty(query('tvae_conv1_x6_bank_bytes', C, R, Cin, k) // 4, device=dev)
call('tvae_bank_split3', bank, a3, a3.numel() * 4, C, R, Cin, k)
out = torch.empty(C, B * R * Ho * Ho, device=dev)
for _ in range(2):
    call('tvae_conv1_fwd_x6', y, a3, bias, out, B, Cin, n, k, pad, C, R, 1, 0.01)
torch.cuda.synchronize()
d = dbg.view(64, 8).cpu()
print('block: prologue  setup  loop(nk)  per-step  epilogue  store-drain  total')
for bk in range(10):
    r = d[bk]
    nk = int(r[6])
    print(bk * 1000, int(r[1] - r[0]), int(r[2] - r[1]), int(r[3] - r[2]), nk, int(r[3] - r[2]) // max(nk, 1), int(r[4] - r[3]), int(r[5] - r[4]), int(r[5] - r[0]))
