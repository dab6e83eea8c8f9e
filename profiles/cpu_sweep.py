"""Thread-count sweep of the CPU baseline (oracle train step) on the GPU box's host cores."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import torch
import bench
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for nt in (8, 16, 32, 64):
    os.environ['TVAE_CPU_THREADS'] = str(nt)
    t0 = time.time()
    r = bench.cpu_baseline(batch=8, steps=1)
    print(nt, r['value'], 'img/s', round(time.time() - t0, 1), 's', flush=True)
