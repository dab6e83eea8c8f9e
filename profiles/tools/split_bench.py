import sys, os, torch
sys.path[:0] = ['/root/repo', '/root/repo/target-vae_amd']
from tvae._lib import call, query
dev = torch.device('cuda:0')
def t(rows, K, tr=0, n=20):
    W = torch.randn(rows, K, device=dev) if not tr else torch.randn(K, rows, device=dev)
    w3 = torch.empty(query('tvae_dense_x6_bytes', rows, K) // 4, device=dev)
    for _ in range(3): call('tvae_dense_split2h', W, W.shape[1], w3, w3.numel() * 4, rows, K, tr, None, None)
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): call('tvae_dense_split2h', W, W.shape[1], w3, w3.numel() * 4, rows, K, tr, None, None)
    e.record(); torch.cuda.synchronize()
    print(rows, K, tr, '%.1f us' % (s.elapsed_time(e) / n * 1e3))
t(49 * 2048, 192); t(512, 512); t(512, 512, 1); t(128, 128)
