#!/usr/bin/env python3
"""Control experiment for stress_determinism.py: the same sharing condition (several processes time-slicing one GPU) with
plain PyTorch operators only -- no kernel of this repository.  Bitwise deviations here would point at the platform
(compute-wave save / restore under preemption), not at this library."""
import sys
import torch
import torch.multiprocessing as mp


def worker(rank, iters, out):
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 16, 44, 44, generator=g).to(dev)
    w = torch.randn(64, 16, 5, 5, generator=g).to(dev)
    big = len(sys.argv) > 3 and sys.argv[3] == 'big'
    a = torch.randn(4096 if big else 2048, 2048 if big else 512, generator=g).to(dev)
    b = torch.randn(2048 if big else 512, 4096 if big else 1024, generator=g).to(dev)
    if big:
        x = torch.randn(32, 16, 96, 96, generator=g).to(dev)
    ref, bad = None, {}
    for it in range(iters):
        y = torch.nn.functional.conv2d(x, w, padding=2)
        z = torch.nn.functional.leaky_relu(a @ b, 0.01)
        s = torch.softmax(z, 1).sum(0)
        t = (y * y).sum((2, 3))
        cur = dict(conv=y, mm=z, softmax=s, red=t)
        if ref is None:
            torch.cuda.synchronize()
            ref = {k: v.clone() for k, v in cur.items()}
            continue
        for k, v in cur.items():
            if not torch.equal(v, ref[k]):
                bad.setdefault(k, []).append((it, int((v != ref[k]).sum())))
    torch.cuda.synchronize()
    with open(out + f'.{rank}', 'w') as f:
        f.write(f'rank {rank}: {iters} iterations, deviations {{k: v[:3] for k, v in bad.items()}} = { {k: v[:3] for k, v in bad.items()} }\n')


if __name__ == '__main__':
    procs, iters = int(sys.argv[1]), int(sys.argv[2])
    mp.start_processes(worker, args=(iters, '/tmp/stress_torch'), nprocs=procs, join=True, start_method='spawn')
    for r in range(procs):
        print(open(f'/tmp/stress_torch.{r}').read(), end='')
