"""Event-timed rotate_bank forward / backward at the bench shapes (gpurun -- python profiles/tools/rotate_bank_probe.py)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'target-vae_amd'))
from tvae import ops

def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dev = torch.device('cuda')
for name, C, Cin, k, R in (('S64', 128, 1, 32, 8), ('S28F', 128, 1, 28, 16), ('S128G', 128, 3, 64, 16)):
    w = torch.randn(C, Cin, 1, k, k, device=dev)
    g = torch.randn(C * R, Cin * k * k, device=dev)
    print('%-6s fwd %7.1f us   bwd %7.1f us   (bank %.1f MB)' % (
        name, t(lambda: ops.rotate_bank(w, R)), t(lambda: ops.rotate_bank_bwd(g, C, Cin, k, R)), g.numel() * 4 / 1e6))
