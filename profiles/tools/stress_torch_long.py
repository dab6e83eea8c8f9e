#!/usr/bin/env python3
"""Control: long-running (>= 1 ms) plain-PyTorch kernels in several processes sharing one GPU, bitwise repeatability."""
import sys
import torch
import torch.multiprocessing as mp


def worker(rank, iters, out):
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1 << 24, generator=g).to(dev)
    m = torch.randn(4096, 4096, generator=g).to(dev)
    img = torch.randn(4, 1, 512, 512, generator=g).to(dev)
    w = torch.randn(8, 1, 31, 31, generator=g).to(dev)
    ref, bad = None, {}
    for it in range(iters):
        cur = dict(cumsum=torch.cumsum(x, 0), sort=torch.sort(x[: 1 << 22])[0], lsm=torch.log_softmax(m, 1),
                   conv=torch.nn.functional.conv2d(img, w, padding=15), mm=m @ m, erf=torch.erf(x).sin().exp().tanh())
        if ref is None:
            torch.cuda.synchronize()
            ref = {k: v.clone() for k, v in cur.items()}
            continue
        for k, v in cur.items():
            if not torch.equal(v, ref[k]):
                bad.setdefault(k, []).append((it, int((v != ref[k]).sum())))
    torch.cuda.synchronize()
    with open(out + f'.{rank}', 'w') as f:
        f.write(f'rank {rank}: {iters} iterations, deviations { {k: v[:3] for k, v in bad.items()} }\n')


if __name__ == '__main__':
    procs, iters = int(sys.argv[1]), int(sys.argv[2])
    mp.start_processes(worker, args=(iters, '/tmp/stress_tl'), nprocs=procs, join=True, start_method='spawn')
    for r in range(procs):
        print(open(f'/tmp/stress_tl.{r}').read(), end='')
