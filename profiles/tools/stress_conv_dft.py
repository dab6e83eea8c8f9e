#!/usr/bin/env python3
"""Stage-level repeatability of the frequency-domain lifting convolution under GPU sharing: calls tvae_conv1_fwd_dft /
tvae_conv1_wgrad_dft on fixed inputs in `procs` concurrent processes and compares, bitwise against the first iteration,
the output AND every intermediate left in the workspace (image spectra A^T, spectral weight W, its split cells W3, T, the
tables), so that a deviation names the stage that produced it.
  python profiles/tools/stress_conv_dft.py <procs> <iters> [B n k pad C R]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import torch
import torch.multiprocessing as mp

PARTS = int(os.environ.get('STRESS_PARTS', '3'))      # 3: exact bf16 split (x6), 2: h3


def worker(rank, iters, geo, out):
    from tvae._lib import call, query
    B, n, k, pad, C, R = geo
    Cin = 1
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(1)
    y = torch.rand(B, Cin, n, n, generator=g).to(dev)
    bank = (torch.randn(C * R, Cin * k * k, generator=g) * (k * k) ** -0.5).to(dev)
    bias = (torch.randn(C, generator=g) * 0.1).to(dev)
    Ho = n + 2 * pad - k + 1
    L, M = n + 2 * pad, C * R
    Lh, K2, NB = L // 2 + 1, 2 * L * Cin, B * Ho
    NBpad = (NB + 127) // 128 * 128
    Mb = (2 * M + 511) // 512 * 512
    w_fl = Lh * Mb * K2
    w3_fl = query('tvae_dense_x6_bytes', Lh * Mb, K2) // 4
    t_fl = Lh * 2 * M * NBpad
    a4 = lambda v: (v + 3) & ~3
    dpre = torch.randn(C, B * R * Ho * Ho, generator=g).to(dev)
    at = torch.zeros(query('tvae_conv1_dft_at_floats', B, Cin, n, k, pad, C, R), device=dev)
    ws = torch.zeros(query('tvae_conv1_dft_ws_floats', B, Cin, n, k, pad, C, R), device=dev)
    outp = torch.empty(C, B * R * Ho * Ho, device=dev)
    dbank = torch.empty(C * R, Cin * k * k, device=dev)
    dbias = torch.empty(C, device=dev)
    # STRESS_SIDE=copy|kernel: a second thread of THIS process keeps another stream busy (device-to-host copies, or
    # elementwise kernels) while the loop runs -- concurrency between hardware queues without a second process
    side = os.environ.get('STRESS_SIDE')
    stop = []
    if side:
        import threading

        def hammer():
            st = torch.cuda.Stream(device=dev)
            big = torch.randn(1 << 22, device=dev)
            host = torch.empty(1 << 22, pin_memory=True)
            with torch.cuda.stream(st):
                while not stop:
                    if side == 'copy':
                        host.copy_(big, non_blocking=True)
                        big.copy_(host, non_blocking=True)
                    else:
                        big.mul_(1.0000001).add_(1e-9)
                    st.synchronize()
        th = threading.Thread(target=hammer, daemon=True)
        th.start()
    ref, bad = None, {}
    for it in range(iters):
        call('tvae_conv1_fwd_dft', y, bank, bias, outp, at, ws, ws.numel(), B, Cin, n, k, pad, C, R, 1, 0.01, PARTS)
        cur = dict(out=outp.clone(), at=at.clone(), W=ws[:w_fl].clone(), W3=ws[a4(w_fl):a4(w_fl) + w3_fl * PARTS // 3].clone(),       # (h3 fills two of the three part arrays)
                   T=ws[a4(w_fl) + a4(w3_fl):a4(w_fl) + a4(w3_fl) + t_fl].clone())
        if os.environ.get('STRESS_FWD_ONLY') != '1':
            call('tvae_conv1_wgrad_dft', dpre, at, dbank, dbias, ws, ws.numel(), B, Cin, n, k, pad, C, R, PARTS)
            cur.update(dbank=dbank.clone(), dbias=dbias.clone(), Sp=ws[:t_fl].clone())
        if ref is None:
            torch.cuda.synchronize()
            ref = cur
            continue
        for nm in cur:
            a, b = cur[nm].view(torch.int32), ref[nm].view(torch.int32)      # bit patterns (NaN-safe)
            if not torch.equal(a, b):
                idx = (a != b).nonzero().flatten()
                ev = (it, int(idx.numel()), int(idx[0]), int(idx[-1]))
                if nm == 'W' and len(bad.get(nm, [])) < 4:      # which (filter, fx, fy, re/im) deviate?
                    fxs = (idx // (Mb * K2))
                    rows = ((idx // K2) % Mb)
                    ks = (idx % K2)
                    ms = (rows % M)
                    desc = []
                    for m_ in ms.unique().tolist():
                        sel = ms == m_
                        fy_ = (ks[sel] % L).unique().tolist()
                        fx_ = fxs[sel].unique().tolist()
                        desc.append((m_, 'fy', fy_, 'fx', fx_, 'rows', rows[sel].unique().tolist(), 'n', int(sel.sum())))
                    ev = ev + (desc,)
                bad.setdefault(nm, []).append(ev)
    torch.cuda.synchronize()
    stop.append(1)
    if side:
        th.join()
    with open(out + f'.{rank}', 'w') as f:
        f.write(f'rank {rank}: geometry {geo} Lh {Lh} M {M} Mb {Mb} K2 {K2} NBpad {NBpad}; {iters} iterations\n')
        for nm, ev in bad.items():
            f.write(f'   {nm}: {len(ev)} deviating iterations (iteration, #words, first, last): {ev[:5]}\n')


if __name__ == '__main__':
    procs, iters = int(sys.argv[1]), int(sys.argv[2])
    geo = tuple(int(v) for v in sys.argv[3:9]) if len(sys.argv) >= 9 else (3, 28, 28, 8, 16, 8)
    mp.start_processes(worker, args=(iters, geo, '/tmp/stress_conv'), nprocs=procs, join=True, start_method='spawn')
    for r in range(procs):
        print(open(f'/tmp/stress_conv.{r}').read(), end='')
