#!/usr/bin/env python3
"""Headline benchmark: TARGET-VAE training images/sec (fwd + bwd + Adam [+ RCCL all-reduce]) on synthetic
64x64 particle stacks, P8 (R=8), z=2, batch 256 per GPU  (BASELINE.json metric / configs[3], SURVEY 8d "S64").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (per-GPU batch fixed).  Rank 0 prints ONE JSON line.  A "step" is one full
training minibatch: encoder fwd, attention head, decoder fwd, likelihood, full backward, gradient all-reduce
(N>1) and the fused Adam update; inputs are resident in HBM before the timed region (like the reference, which
keeps the dataset on the device, train_mnist.py:495).  Noise (Exp(1), N(0,1)) is drawn on the device inside the
step, as the reference does.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'target-vae_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)



def visible_gpu_count() -> int:
    """GPUs this process may use, WITHOUT any HIP / torch call (the spawning parent must stay GPU-free: a process that
    has initialised the GPU must not be replaced or forked into ranks): the KFD topology lists one node per agent and
    GPU agents are the nodes with simd_count > 0; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    restrict that set when present."""
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None and v.strip() != '':
            return len([t for t in v.split(',') if t.strip() != ''])
    base = '/sys/class/kfd/kfd/topology/nodes'
    n = 0
    try:
        for node in os.listdir(base):
            try:
                props = open(os.path.join(base, node, 'properties')).read()
            except OSError:
                continue
            for line in props.splitlines():
                f = line.split()
                if len(f) == 2 and f[0] == 'simd_count' and int(f[1]) > 0:
                    n += 1
    except OSError:
        return 0
    return n


def spawn_ranks(n: int, timeout_s: float = 3600.0) -> int:
    """`python bench.py --gpus N` without a torchrun environment.  This parent never touches the GPU and never imports
    torch (it runs BEFORE the module-level `import torch` below; devices are counted from sysfs / the *_VISIBLE_DEVICES
    variables); it starts one child per GPU with the torchrun variables (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*),
    lets rank 0 print the JSON line and POLLS the children: when one exits non-zero (RCCL init failure, out of memory)
    the others -- which would otherwise block forever in their next collective -- are terminated and that exit code is
    returned; the same after `timeout_s`."""
    import socket
    import subprocess
    have = visible_gpu_count()
    # TVAE_BENCH_REHEARSE=1: a dress rehearsal of the N-rank code path on ONE GPU -- every rank on device 0, collectives over
    # gloo (RCCL refuses two ranks on one device).  It exists so that everything that only runs with WORLD_SIZE > 1 (replica
    # broadcast, early bucket, diagnostics, strong-scaling block) has executed before the driver's first multi-GPU run; its
    # numbers mean nothing.  At most a handful of ranks (the GPU boxes allow few processes on the card).
    rehearse = os.environ.get('TVAE_BENCH_REHEARSE', '0') == '1'
    if have < n and not (rehearse and have >= 1 and n <= 4):
        print(f'bench.py: --gpus {n} but only {have} GPU(s) are visible', file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if rehearse else r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        if rehearse:
            env['TVAE_DP_BACKEND'] = 'gloo'
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    t_end = time.monotonic() + timeout_s
    rc = 0
    live = list(procs)
    while live:
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
        if live and (rc != 0 or time.monotonic() > t_end):
            if rc == 0:
                rc = 124
                print(f'bench.py: ranks still running after {timeout_s:.0f} s, terminating them', file=sys.stderr)
            for pr in live:
                pr.terminate()
            for pr in live:
                try:
                    pr.wait(timeout=15)
                except subprocess.TimeoutExpired:
                    pr.kill()
                    pr.wait()
            break
        if live:
            time.sleep(0.2)
    return rc


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (BASELINE: 256; S128G: 8)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-f32-companion', action='store_true',
                    help="skip the extra measurements of the same steps in the all-fp32-MFMA mode (TVAE_GEMM=f32) and in "
                         "the opt-in bf16 throughput mode")
    ap.add_argument('--no-small-batch', action='store_true',
                    help='skip the 32 / 64 / 128 images-per-step points and the one-rank all-reduce timing (N = 1)')
    ap.add_argument('--no-workloads', action='store_true',
                    help='skip the S28 / S28F / S128G companion measurements of the default run (N = 1)')
    ap.add_argument('--no-strong', action='store_true',
                    help='N > 1: skip the extra strong-scaling measurement (global batch fixed at the per-GPU batch)')
    ap.add_argument('--graph', action='store_true',
                    help='replay a captured hipGraph of forward + backward (tvae/graph.py: bitwise the eager result) in the '
                         'timed loop; without the flag the 28x28 workloads report it as a companion measurement')
    ap.add_argument('--workload', choices=['S128G', 'S28', 'S28F', 'S64'], default='S64',
                    help='S64 = the BASELINE.json metric configuration (default); others are extra measurements')
    return ap.parse_args()


if __name__ == '__main__' and 'WORLD_SIZE' not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:                       # become the GPU-free parent of N ranks before torch is even imported
        sys.exit(spawn_ranks(_a.gpus))

import numpy as np
import torch
import torch.distributed as dist

# /opt/skills/guides/MI355X_MICROARCH.md: dense matrix peaks.  In the default 'x6' arithmetic the lifting convolution
# runs SIX v_mfma_f32_32x32x16_bf16 per algorithmic product block (exact 3 x bf16 operand split, fp32-equivalent
# result), so its speed of light in algorithmic FLOP/s is the bf16 peak / 6.
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
MODE_INFO = {
    'f32': dict(peak=PEAK_F32_MFMA_TFLOPS, insn='v_mfma_f32_32x32x2_f32', suffix='',
                kernels={'tvae_conv1_fwd': 'conv1_fwd_img_kernel<true,2>',
                         'tvae_conv1_wgrad': 'conv1_wgrad_img_kernel<1,32,2>'},
                dtype='f32'),
    'x6': dict(peak=PEAK_BF16_MFMA_TFLOPS / 6.0, insn='6 x v_mfma_f32_32x32x16_bf16 per product block', suffix='_x6',
               kernels={'tvae_conv1_fwd': 'conv1_fwd_x6_kernel', 'tvae_conv1_wgrad': 'conv1_wgrad_x6_kernel'},
               dtype='f32 (matrix products: operands split exactly into 3 x bf16, 6 bf16 MFMAs per product, fp32 '
                     'accumulate -- fp32-equivalent, same parity tolerances; the 512-wide decoder layers and the forward / data '
                     'gradient of the 128-wide encoder 1x1x1 layers likewise, their weight gradient on the fp32 MFMA)'),
    'h3': dict(peak=PEAK_BF16_MFMA_TFLOPS / 3.0, insn='3 x v_mfma_f32_32x32x16_f16 per product block', suffix='_x6',
               kernels={'tvae_conv1_fwd': 'conv1_fwd_x6_kernel', 'tvae_conv1_wgrad': 'conv1_wgrad_x6_kernel'},
               dtype='f32 (matrix products: operands as TWO fp16 parts under one power-of-two scale PER ROW of the operand '
                     '(row / column of the product), 3 fp16 MFMAs per product block -- 2 where one operand is the exact 0 / 1 '
                     'matrix -- fp32 accumulate: at least as accurate against fp64 as the fp32 matrix pipe '
                     '(profiles/experiments/f16_split_probe.hip), rows 2^-32 below their tensor to 1e-5 of themselves '
                     '(tests/test_hip_primitives.py::test_h3_row_dynamic_range_*), same parity tolerances; launches without '
                     'an h3 instance run the exact 3 x bf16 split and are priced as such: roofline.dense_launches.'
                     'mfma_per_block_as_launched)'),
    # opt-in throughput mode (TVAE_GEMM=bf16; BASELINE.json configs 2 / 5): NOT the headline, not fp32-equivalent
    'bf16': dict(peak=PEAK_BF16_MFMA_TFLOPS, insn='1 x v_mfma_f32_32x32x16_bf16 per product block', suffix='_x6',
                 kernels={'tvae_conv1_fwd': 'conv1_fwd_x6_kernel', 'tvae_conv1_wgrad': 'conv1_wgrad_x6_kernel'},
                 dtype='bf16 (matrix products of the convolution and the 512-wide decoder layers: operands rounded to one '
                       'bf16 number, fp32 accumulate -- the opt-in throughput mode, tolerance 2e-2 on the ELBO; the weight '
                       'gradient of the 128-wide encoder 1x1x1 layers on the fp32 MFMA)'),
}


def pmc_traffic(entry, grid=None, kernel=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/pmc_traffic.json: FETCH_SIZE x2 -- gfx950 reports half the fetched bytes, calibrated on outer_mask --
    plus WRITE_SIZE).  Counters cannot be collected inside the timed run, so this is the committed measurement; `kernel`
    (the kernel family the bench attaches it to) must be the one the counters were collected on, else None: a stale file
    must not decorate a different launch."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        e = d[entry + '@grid%d' % grid] if (grid is not None and (entry + '@grid%d' % grid) in d) else d[entry]
        if kernel is not None and kernel not in e.get('kernel', ''):
            print(f'# bench.py: profiles/pmc_traffic.json[{entry}] was collected on {e.get("kernel")!r}, not on {kernel!r}: '
                  'traffic omitted (re-run profiles/collect.sh)', file=sys.stderr)
            return None
        return e['hbm_bytes_per_launch']
    except Exception:
        return None
CFG = dict(n=64, cin=1, zd=2, C=128, k=64, pad=16, R=8, hidden=512, layers=2, n_out=1)
# optional extra workloads (BASELINE.json configs[1], configs[2]); the default S64 is the metric's configuration
WORKLOADS = {
    'S64': dict(CFG, fourier=False, lik='gauss', data='randn',
                desc='S64: synthetic 64x64 particle stack (torch.randn), P8 group-conv encoder k=64 p=16 C=128, z=2, '
                     't-inf attention, r-inf attention+offsets, decoder 2->512->512->1, Gaussian likelihood, Adam lr 2e-4'),
    'S28': dict(n=28, cin=1, zd=2, C=128, k=28, pad=8, R=8, hidden=512, layers=2, n_out=1, fourier=False, lik='bce',
                data='rand', desc='S28: synthetic 28x28 MNIST-shape stack (torch.rand), P8 k=28 p=8 C=128, z=2, '
                                  'decoder 2->512->512->1, BCE likelihood'),
    'S28F': dict(n=28, cin=1, zd=2, C=128, k=28, pad=8, R=16, hidden=512, layers=2, n_out=1, fourier=True, lik='bce',
                 data='rand', desc='S28F: synthetic 28x28, P16 k=28 p=8 C=128, z=2, Fourier decoder '
                                   'cos1024->512->512->1, BCE likelihood'),
    # BASELINE.json configs[4] (galaxy shape): 136 MB of lifted activations per image, so 8 images per step
    'S128G': dict(n=128, cin=3, zd=50, C=128, k=64, pad=32, R=16, hidden=512, layers=4, n_out=3, fourier=True, lik='bce3',
                  data='rand', batch=8,
                  desc='S128G: synthetic 128x128x3 galaxy-shape stack (torch.rand), P16 k=64 p=32 C=128, z=50, Fourier '
                       'decoder cos1024->512->512->512->512->3, 3-channel BCE likelihood'),
}


def conv1_flops_per_image(c=CFG):
    """ALGORITHMIC FLOPs of the lifting convolution per image: 2*C*R*Cin*k^2*Ho^2."""
    ho = c['n'] + 2 * c['pad'] - c['k'] + 1
    return 2.0 * c['C'] * c['R'] * c['cin'] * c['k'] ** 2 * ho ** 2


def build_models(device, c=None):
    import src.models as M
    torch.manual_seed(0)            # reference default init; generator constructed first (train_mnist.py:522,551)
    c = c or WORKLOADS['S64']
    gen = M.SpatialGenerator(c['zd'], c['hidden'], n_out=c['n_out'], num_layers=c['layers'],
                             fourier_expansion=c.get('fourier', False), sigma=2.0 / (c['n'] - 1))
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        c['n'], c['cin'], c['zd'], kernels_num=c['C'], kernels_size=c['k'], padding=c['pad'], groupconv=c['R'],
        rot_refinement=True, theta_prior=np.pi, normal_prior_over_r=False)
    return gen.to(device), enc.to(device)


def cpu_baseline(batch=32, steps=4):
    """The CPU oracle (a port of the reference step onto the same ATen CPU operators) timed on this host's
    cores on a bounded sample of the same workload: `batch` images per step, 1 warm-up + `steps` timed steps."""
    from oracle import tvae_oracle as O
    import src.models as M
    torch.manual_seed(0)
    c = CFG
    gen = M.SpatialGenerator(c['zd'], c['hidden'], n_out=c['n_out'], num_layers=c['layers'])
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        c['n'], c['cin'], c['zd'], kernels_num=c['C'], kernels_size=c['k'], padding=c['pad'], groupconv=c['R'],
        rot_refinement=True, theta_prior=np.pi, normal_prior_over_r=False)
    encp = {k_: v.detach().clone().requires_grad_(True) for k_, v in enc.state_dict().items()}
    genp = {k_: v.detach().clone().requires_grad_(True) for k_, v in gen.state_dict().items()}
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    # thread sweep on the GPU box's host (profiles/cpu_sweep.py): 8/16/32/64 threads -> 15/22/17/11 img/s
    cores = int(os.environ.get('TVAE_CPU_THREADS', min(avail, 16)))
    torch.set_num_threads(cores)
    st = O.new_opt_state(encp, genp)
    x = O.image_coords(c['n'])
    ho = c['n'] + 2 * c['pad'] - c['k'] + 1
    ts = []
    for i in range(steps + 1):
        torch.manual_seed(100 + i)
        y = torch.randn(batch, c['cin'], c['n'], c['n'])
        noise = dict(E=torch.empty(batch, c['R'] * ho * ho).exponential_(), eps_z=torch.randn(batch, c['zd']),
                     eps_theta=torch.randn(batch))
        t0 = time.perf_counter()
        O.train_step(x, y, encp, genp, st, noise, R=c['R'], padding=c['pad'], rot_refinement=True,
                     theta_prior=np.pi, normal_prior_over_r=False, num_layers=c['layers'], likelihood='gauss')
        ts.append(time.perf_counter() - t0)
    best = min(ts[1:])
    # encoder forward only (the north-star roofline target is stated on it): same oracle, no gradients
    te = []
    sd = {k_: v.detach() for k_, v in encp.items()}
    with torch.no_grad():
        for i in range(3):
            t0 = time.perf_counter()
            O.encoder_forward(sd, y, noise['E'], c['R'], c['pad'], True, np.pi, False)
            te.append(time.perf_counter() - t0)
    return dict(value=batch / best, unit='images/sec', cores=cores, kind='port',
                sample=f'oracle train step (fwd+bwd+Adam), {batch} images/step, 1 warm-up + {steps} timed steps, '
                       f'min; torch {torch.__version__} CPU, {cores} threads',
                encoder_forward={'value': batch / min(te[1:]), 'unit': 'images/sec',
                                 'sample': f'oracle encoder forward, {batch} images, 1 warm-up + 2 timed, min'})


def measure_extra_workload(name, dev, steps, warmup=2):
    """One of the other BASELINE.json workloads (configs[1], [2], [4]) inside the default, driver-timed run: the same
    training step (fwd + bwd + Adam, noise drawn on the device, data resident in HBM) at that workload's own batch, timed with
    the bench's barrier bracket.  A few steps only (~0.5 s of GPU time for all three): these are companion numbers so that
    the 28x28 / galaxy-shape figures of the north star do not exist only in builder-run profiles."""
    from tvae import optim, step
    # 'S128G@32': the galaxy workload at 32 images per step next to its 8 (136 MB of lifted activations per
    # image is 4.4 GB at 32 -- nothing in 288 GB -- and the per-launch fixed costs are shared by four times the images)
    name, _, b_over = name.partition('@')
    c = WORKLOADS[name]
    B = int(b_over) if b_over else c.get('batch', 256)
    gen, enc = build_models(dev, c)
    opt = optim.FlatAdam(list(gen.parameters()) + list(enc.parameters()), lr=2e-4)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    mk = torch.randn if c['data'] == 'randn' else torch.rand
    data = mk(2 * B, c['cin'], c['n'], c['n'], device=dev, generator=g)
    x = torch.from_numpy(__import__('tvae.tables', fromlist=['x']).image_coords(c['n'])).to(dev)
    step.pixel_spacing(x)

    def one(i):
        y = data[(i % 2) * B:(i % 2) * B + B]
        elbo, _, _ = step.elbo_terms(x, y, gen, enc, c['lik'])
        step.backward_neg_elbo(elbo)         # = (-elbo).backward() of the training loop (tvae/step.py)
        opt.step()
        opt.zero_grad(set_to_none=True)
        return elbo.detach()

    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = one(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'workload': c['desc'], 'images_per_step': B, 'steps': steps, 'warmup': warmup,
            'ms_per_step': 1e3 * dt / steps, 'value': B * steps / dt, 'unit': 'images/sec', 'elbo': float(last)}


def main():
    args = parse_args()
    assert sorted(WORKLOADS) == ['S128G', 'S28', 'S28F', 'S64']

    from tvae import dp, ops, optim, step
    from tvae import _lib
    rank, world, local = dp.init_from_env(backend=os.environ.get('TVAE_DP_BACKEND') or None)
    if world != args.gpus:
        if rank == 0:
            print(f'# note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE', file=sys.stderr)
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    wl = WORKLOADS[args.workload]
    if args.batch is None:
        args.batch = wl.get('batch', 256)
    gen, enc = build_models(dev, wl)
    params = list(gen.parameters()) + list(enc.parameters())
    reducer = dp.GradReducer() if world > 1 else None
    opt = optim.FlatAdam(params, lr=2e-4, reducer=reducer, early_params=len(list(gen.parameters())))
    if world > 1:                                   # identical replicas: rank 0's flat parameters and every buffer
        dist.broadcast(opt.flat_p, src=0)
        dp.broadcast_buffers(gen, enc)

    B, c = args.batch, wl
    total_steps = args.steps + args.warmup
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    # synthetic dataset resident in HBM: torch.randn matches --normalize'd particles (SURVEY 8d "S64")
    n_img = B * min(total_steps, 8)
    mk = torch.randn if c['data'] == 'randn' else torch.rand
    data = mk(n_img, c['cin'], c['n'], c['n'], device=dev, generator=g)
    x = torch.from_numpy(__import__('tvae.tables', fromlist=['x']).image_coords(c['n'])).to(dev)
    step.pixel_spacing(x)                           # cached once (the reference syncs for it every step)

    gs_box = [None]
    if args.graph:
        from tvae import graph as _graph
        gs_box[0] = _graph.GraphedStep(x, gen, enc, opt, c['lik'], B, (c['cin'], c['n'], c['n']), dev)

    def one_step(i, b=None):
        b = B if b is None else b
        lo = (i % (n_img // B)) * B
        y = data[lo:lo + b]
        if gs_box[0] is not None and b == gs_box[0].B:      # captured forward + backward; noise, all-reduce, Adam outside
            terms = gs_box[0].run(y)
            opt.step()
            e_ = terms[0].clone()
            opt.zero_grad()
            return e_
        elbo, log_p, kl = step.elbo_terms(x, y, gen, enc, c['lik'])
        step.backward_neg_elbo(elbo)         # = (-elbo).backward() of the training loop (tvae/step.py)
        opt.step()
        opt.zero_grad(set_to_none=True)      # as the training loop (tvae/step.py): gradients gathered by one multi-tensor copy
        return elbo.detach()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    mode = _lib.get_gemm_mode()
    for i in range(args.warmup):
        one_step(i)
    barrier()
    ops.KERNEL_EVENTS = {}
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = one_step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    early_posted_main = reducer.posted_early if reducer is not None else 0      # (warm-up + timed steps; diagnostics below)
    kev = ops.kernel_event_ms()
    ops.KERNEL_EVENTS = None
    if gs_box[0] is not None:
        # --graph: the replayed launches carry no per-entry-point events; the roofline block comes from the same number of EAGER
        # steps timed right after (the headline `value` stays the graph-replay run)
        g_saved, gs_box[0] = gs_box[0], None
        for i in range(2):
            one_step(i)
        barrier()
        ops.KERNEL_EVENTS = {}
        for i in range(args.steps):
            one_step(args.warmup + i)
        barrier()
        kev = ops.kernel_event_ms()
        ops.KERNEL_EVENTS = None
        gs_box[0] = g_saved
    # N > 1: the same number of steps again with the GLOBAL batch fixed at the per-GPU batch of the weak run (strong
    # scaling, SURVEY 8d: "B=256 per GPU (weak) and global-B=256 (strong) separately and label them")
    strong = None
    if world > 1 and not args.no_strong and B >= world:
        bs = B // world
        for i in range(2):
            one_step(i, bs)
        barrier()
        ts0 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i, bs)
        barrier()
        dts = time.perf_counter() - ts0
        tt = torch.tensor([dts], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dts = float(tt.item())
        strong = {'scaling': 'strong', 'global_batch': bs * world, 'per_gpu_batch': bs,
                  'value': world * bs * args.steps / dts, 'unit': 'images/sec', 'ms_per_step': 1e3 * dts / args.steps,
                  'steps': args.steps, 'warmup': 2}
    # single-GPU proxy for the strong-scaling question: what does a step cost when the SAME global batch of
    # 256 is divided over 8 / 4 / 2 GPUs, i.e. 32 / 64 / 128 images per step on one GPU -- ms per step, images/s and the ratio
    # of the per-image rate to the B = 256 rate (1.0 = every grid still fills the chip).  Plus the measured duration of the
    # gradient all-reduce's one-rank RCCL call on the real 3.2 MB flat buffer (launch + kernel; the multi-rank ring time is the
    # driver's to measure).  If 32 images per GPU run well under 75 % of the B = 256 rate, 6x at 8 GPUs is out of reach for
    # the strong-scaling line whatever the collective costs.
    small_batch = None
    if world == 1 and args.workload == 'S64' and not (args.no_small_batch or args.no_f32_companion) and gs_box[0] is None:
        small_batch = {'per_image_rate_at_full_batch': B * args.steps / dt, 'full_batch': B, 'points': []}
        for bs in (12, 32, 64, 100, 128):
            if bs >= B:
                continue
            for i in range(2):
                one_step(i, bs)
            barrier()
            tsb = time.perf_counter()
            for i in range(args.steps):
                one_step(args.warmup + i, bs)
            barrier()
            dsb = time.perf_counter() - tsb
            # ... and as the training CLIs run it since round 6: forward + backward replayed from a captured hipGraph
            # (tvae/graph.py; bitwise the eager step).  The launch count, not the arithmetic, is what a step of this size costs.
            g_ms = None
            if bs in (12, 32):
                try:
                    from tvae import graph as _graph
                    gsb = _graph.GraphedStep(x, gen, enc, opt, c['lik'], bs, (c['cin'], c['n'], c['n']), dev)
                    saved_g, gs_box[0] = gs_box[0], gsb
                    for i in range(2):
                        one_step(i, bs)
                    barrier()
                    tg = time.perf_counter()
                    for i in range(args.steps):
                        one_step(args.warmup + i, bs)
                    barrier()
                    g_ms = 1e3 * (time.perf_counter() - tg) / args.steps
                    gs_box[0] = saved_g
                    gsb.close()
                    opt.zero_grad(set_to_none=True)
                except Exception as ex:                  # a companion must never take the headline down
                    g_ms = repr(ex)[:200]
            small_batch['points'].append({'images_per_step': bs, 'ms_per_step': 1e3 * dsb / args.steps,
                                          'ms_per_step_graph_replay': g_ms,
                                          'value': bs * args.steps / dsb, 'unit': 'images/sec',
                                          'ratio_to_full_batch_rate': (bs * args.steps / dsb) / (B * args.steps / dt),
                                          **({'implied_speedup_at_%d_gpus' % (B // bs): (B // bs) * (bs * args.steps / dsb) /
                                              (B * args.steps / dt)} if B % bs == 0 else
                                             {'note': "the reference's default --minibatch-size 100 (train_mnist.py:426) on one "
                                                      'rank' if bs == 100 else 'the per-rank share of that default minibatch on 8 '
                                                      'ranks (13 / 12 images)'})})
        try:
            if not dist.is_initialized():
                import tempfile
                store = tempfile.NamedTemporaryFile(prefix='tvae_bench_pg_', delete=True)
                dist.init_process_group('nccl', init_method='file://' + store.name + '.store', rank=0, world_size=1)
                made_pg = True
            else:
                made_pg = False
            buf = torch.zeros_like(opt.flat_g)
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            for _ in range(20):
                dist.all_reduce(buf)
            eb.record()
            torch.cuda.synchronize()
            small_batch['allreduce_one_rank'] = {'backend': dist.get_backend() + ' (RCCL)', 'payload_bytes': int(buf.numel()) * 4,
                                                 'ms_per_call': ea.elapsed_time(eb) / 20,
                                                 'note': 'world_size 1: launch + the collective kernel on the real gradient '
                                                         'buffer; not a ring time'}
            if made_pg:
                dist.destroy_process_group()
                try:
                    os.unlink(store.name + '.store')      # the file:// rendezvous (do not leave it behind)
                except OSError:
                    pass
        except Exception as ex:       # the collective is a side measurement: never let it take the bench line down
            small_batch['allreduce_one_rank'] = {'error': repr(ex)[:200]}
    # companion measurement: the same number of steps with every matrix product on the exact fp32 MFMA
    companion = None
    if world == 1 and mode in ('x6', 'h3') and not args.no_f32_companion:
        _lib.set_gemm_mode('f32')
        for i in range(2):
            one_step(i)
        barrier()
        ops.KERNEL_EVENTS = {}
        t1 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i)
        barrier()
        dt1 = time.perf_counter() - t1
        kev1 = ops.kernel_event_ms()
        ops.KERNEL_EVENTS = None
        _lib.set_gemm_mode(mode)
        fl1 = conv1_flops_per_image(c) * B
        companion = {'value': B * args.steps / dt1, 'unit': 'images/sec', 'ms_per_step': 1e3 * dt1 / args.steps,
                     'arithmetic': 'TVAE_GEMM=f32: every matrix product on v_mfma_f32_32x32x2_f32',
                     'conv1_fwd_ms': kev1.get('tvae_conv1_fwd', {}).get('mean_ms'),
                     'conv1_wgrad_ms': kev1.get('tvae_conv1_wgrad', {}).get('mean_ms'),
                     'conv1_wgrad_frac_of_f32_peak':
                         fl1 / (kev1['tvae_conv1_wgrad']['mean_ms'] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS
                         if 'tvae_conv1_wgrad' in kev1 else None}
    # companion: the exact three-part bf16 split (six products per block; the default arithmetic of rounds 1-2)
    companion_x6 = None
    if world == 1 and mode == 'h3' and not args.no_f32_companion:
        _lib.set_gemm_mode('x6')
        for i in range(2):
            one_step(i)
        barrier()
        ops.KERNEL_EVENTS = {}
        t3 = time.perf_counter()
        lastx = None
        for i in range(args.steps):
            lastx = one_step(args.warmup + i)
        barrier()
        dt3 = time.perf_counter() - t3
        kev3 = ops.kernel_event_ms()
        ops.KERNEL_EVENTS = None
        _lib.set_gemm_mode(mode)
        companion_x6 = {'value': B * args.steps / dt3, 'unit': 'images/sec', 'ms_per_step': 1e3 * dt3 / args.steps,
                        'arithmetic': 'TVAE_GEMM=x6: operands split exactly into 3 x bf16, 6 MFMAs per product block '
                                      '(3 against the 0 / 1 operand)',
                        'elbo': float(lastx),
                        'entry_points_ms': {k_: round(v['mean_ms'], 3) for k_, v in sorted(kev3.items())
                                            if k_.startswith('tvae_linear') or k_.startswith('tvae_conv1')}}
    # second companion: the opt-in bf16 throughput mode (one bf16 MFMA per product block; never the headline)
    companion_bf16 = None
    if world == 1 and mode in ('x6', 'h3') and not args.no_f32_companion:
        _lib.set_gemm_mode('bf16')
        for i in range(2):
            one_step(i)
        barrier()
        ops.KERNEL_EVENTS = {}
        t2 = time.perf_counter()
        lastb = None
        for i in range(args.steps):
            lastb = one_step(args.warmup + i)
        barrier()
        dt2 = time.perf_counter() - t2
        kev2 = ops.kernel_event_ms()
        ops.KERNEL_EVENTS = None
        _lib.set_gemm_mode(mode)
        companion_bf16 = {'value': B * args.steps / dt2, 'unit': 'images/sec', 'ms_per_step': 1e3 * dt2 / args.steps,
                          'arithmetic': 'TVAE_GEMM=bf16: operands of the convolution and decoder GEMMs rounded to one bf16 '
                                        'number, fp32 accumulate; since round 4 the two large intermediates of the frequency-'
                                        'domain convolution, T and S\', are also STORED as bf16 '
                                        '(tolerance 2e-2 on the ELBO terms, tests/test_hip_modules.py::test_bf16_throughput_mode; '
                                        '1.5e-2 on the convolution, tests/test_hip_primitives.py::test_conv1_dft_bf16_mode); not '
                                        'fp32-equivalent, not the headline',
                          'elbo': float(lastb),
                          'entry_points_ms': {k_: round(v['mean_ms'], 3) for k_, v in sorted(kev2.items())
                                              if k_.startswith('tvae_linear') or k_.startswith('tvae_conv1')}}
    # third companion (28x28 workloads, where ~70 launches per 4 ms step make the host's launch cost visible): the same steps
    # with forward + backward replayed from a hipGraph (tvae/graph.py; bitwise the eager result)
    companion_graph = None
    if world == 1 and mode in ('x6', 'h3') and not args.graph and not args.no_f32_companion and args.workload in ('S28', 'S28F'):
        from tvae import graph as _graph
        gs_box[0] = _graph.GraphedStep(x, gen, enc, opt, c['lik'], B, (c['cin'], c['n'], c['n']), dev)
        for i in range(2):
            one_step(i)
        barrier()
        t3 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i)
        barrier()
        dt3 = time.perf_counter() - t3
        gs_box[0] = None
        companion_graph = {'value': B * args.steps / dt3, 'unit': 'images/sec', 'ms_per_step': 1e3 * dt3 / args.steps,
                           'what': 'forward + backward replayed from a captured hipGraph (--graph); noise draws, Adam and the '
                                   'minibatch copy stay eager'}
    # encoder forward only (SURVEY 8d "Metric"; BASELINE north_star states its roofline target on it): the same 256
    # images through conv1 -> conv2 -> heads -> attention head in training mode (the two activations a backward needs are
    # written), timed with events on the launch stream
    enc_fwd = enc_inf = None
    if world == 1 and args.workload == 'S64':
        ho = c['n'] + 2 * c['pad'] - c['k'] + 1
        yb = data[:B]
        P = c['R'] * ho * ho

        def time_encoder(training):
            # training = True: grad mode on, the forward keeps what a backward needs (conv1 / conv2 activations, sign words,
            # image spectra); False: under torch.no_grad() like eval_model (train_mnist.py:352-387) and get_latent
            # (clustering_mnist.py:121-161) -- the inference-mode kernels (tvae/ops.py: no H, no sign words)
            with torch.enable_grad() if training else torch.no_grad():
                for _ in range(2):
                    enc(yb, dev)
                torch.cuda.synchronize()
                ops.KERNEL_EVENTS = {}
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    enc(yb, dev)
                e1.record()
                torch.cuda.synchronize()
                kev_e = ops.kernel_event_ms()
                ops.KERNEL_EVENTS = None
            return e0.elapsed_time(e1) / args.steps, kev_e

        ms_e, kev_e = time_encoder(True)
        # algorithmic bytes per image, training mode: input + heads + both saved activations + bank / B
        bytes_img = 4.0 * (c['cin'] * c['n'] ** 2 + (3 + 2 * c['zd']) * P + 2 * c['C'] * P) + \
            4.0 * c['C'] * c['R'] * c['cin'] * c['k'] ** 2 / B
        fl_img = conv1_flops_per_image(c) + 2.0 * c['C'] * c['C'] * P + 2.0 * c['C'] * (3 + 2 * c['zd']) * P
        enc_fwd = {'value': B / (ms_e * 1e-3), 'unit': 'images/sec', 'ms': ms_e, 'mode': 'training (grad enabled)',
                   'algorithmic_bytes_per_image': bytes_img,
                   'hbm': {'achieved': bytes_img * B / (ms_e * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                           'frac': bytes_img * B / (ms_e * 1e-3) / 8.0e12},
                   'mfma': {'direct_form_tflops': fl_img * B / (ms_e * 1e-3) / 1e12,
                            'note': 'direct-form FLOPs (2*C*R*k^2*Ho^2 + 1x1x1 layers) over the measured time; the '
                                    'frequency-domain convolution executes 13x fewer, so this exceeds every peak'},
                   'conv1_fwd_ms': kev_e.get('tvae_conv1_fwd', {}).get('mean_ms'),
                   'enc_tail_fwd_ms': kev_e.get('tvae_enc_tail_fwd_x6', {}).get('mean_ms')}
        ms_i, kev_i = time_encoder(False)
        # SURVEY 8d byte figure (i), the module-boundary minimum of a fully fused inference encoder: input + the 7 encoder
        # outputs (3 + 2 + 2z rows of P positions) + the rotated bank amortised over the batch = 0.396 MB / image at cfg4
        bytes_inf = 4.0 * (c['cin'] * c['n'] ** 2 + (3 + 2 + 2 * c['zd']) * P) + \
            4.0 * c['C'] * c['R'] * c['cin'] * c['k'] ** 2 / B
        enc_inf = {'value': B / (ms_i * 1e-3), 'unit': 'images/sec', 'ms': ms_i,
                   'mode': 'inference (torch.no_grad(): no conv2 activation, no sign words, nothing retained)',
                   'algorithmic_bytes_per_image': bytes_inf,
                   'hbm': {'achieved': bytes_inf * B / (ms_i * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                           'frac': bytes_inf * B / (ms_i * 1e-3) / 8.0e12},
                   'conv1_fwd_ms': kev_i.get('tvae_conv1_fwd', {}).get('mean_ms'),
                   'enc_tail_fwd_ms': kev_i.get('tvae_enc_tail_fwd_x6', {}).get('mean_ms'),
                   'speedup_over_training_forward': ms_e / ms_i}
    # the other BASELINE.json workloads in the same driver-timed line: S28 (configs[1]), S28F (configs[2]),
    # S128G (configs[4]) at their own batch, default arithmetic
    workloads = None
    if world == 1 and args.workload == 'S64' and not args.no_workloads and gs_box[0] is None:
        workloads = {}
        for wn in ('S28', 'S28F', 'S128G', 'S128G@32'):
            try:
                workloads[wn] = measure_extra_workload(wn, dev, min(args.steps, 10))
            except Exception as ex:          # a companion must never take the headline down
                workloads[wn] = {'error': repr(ex)[:300]}
            torch.cuda.empty_cache()
    dp_diag = None
    if world > 1:
        # self-diagnosis of the first multi-GPU run: every rank's own wall time of the timed region, the
        # early buckets it posted from inside a backward, the collectives it issued -- gathered, so that a straggler, a rank
        # that never posted the early bucket or a group of the wrong size is visible in the one JSON line
        mine = torch.tensor([dt, float(early_posted_main), float(dist.get_world_size()), float(torch.cuda.current_device())],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per = [[float(v) for v in t_.tolist()] for t_ in allr]
        dp_diag = {'per_rank_ms_per_step': [1e3 * p_[0] / args.steps for p_ in per],
                   'ms_per_step_min': 1e3 * min(p_[0] for p_ in per) / args.steps,
                   'ms_per_step_max': 1e3 * max(p_[0] for p_ in per) / args.steps,
                   'early_buckets_posted_per_rank': [int(p_[1]) for p_ in per],
                   'early_buckets_expected': (args.warmup + args.steps) if opt._early_n else 0,
                   'world_size_seen_per_rank': [int(p_[2]) for p_ in per],
                   'device_index_per_rank': [int(p_[3]) for p_ in per],
                   'note': 'steps of the warm-up and the timed region both post the early (decoder) bucket'}
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    elbo_last = float(last)

    if rank == 0:
        imgs = world * B * args.steps
        conv_flops = conv1_flops_per_image(c) * B
        Nt = B * c['n'] * c['n']
        dense_flops = 2.0 * c['hidden'] * c['hidden'] * Nt
        info = MODE_INFO[mode]
        # every timed entry point: (algorithmic FLOPs per launch, kernel, note)
        conv_dft = mode in ('x6', 'h3', 'bf16') and bool(int(os.environ.get('TVAE_CONV_DFT', '1')))
        entries = {
            'tvae_conv1_fwd': (conv_flops, 'dft_spectra_x_kernel + batched dense_x6_xres_kernel (other frames: _plain4_kernel) + dft_out_ring_kernel'
                               if conv_dft else info['kernels']['tvae_conv1_fwd']),
            'tvae_conv1_wgrad': (conv_flops, 'dft_dy_ring_kernel + batched dense_wgrad_x6_dma_kernel (512 x 128 tile) + dft_dbank_x_kernel'
                                 if conv_dft else info['kernels']['tvae_conv1_wgrad']),
            'tvae_linear_fwd_x6': (dense_flops, 'dense_x6_kernel'),
            'tvae_linear_dgrad_x6': (dense_flops, 'dense_x6_kernel'),
            'tvae_linear_wgrad_x6': (dense_flops, 'dense_wgrad_x6_dma_kernel'),
        }
        # fused encoder tail (conv2 + head projection; csrc/enc_tail_x6_kernels.hpp): HBM-bound side kernels
        ho_ = c['n'] + 2 * c['pad'] - c['k'] + 1
        ncol = B * c['R'] * ho_ * ho_
        nh_ = 3 + 2 * c['zd']
        tail_flops = 2.0 * c['C'] * (c['C'] + nh_) * ncol
        entries['tvae_enc_tail_fwd_x6'] = (tail_flops, 'enc_tail_fwd_x6_kernel')
        entries['tvae_enc_tail_dgrad_x6'] = (tail_flops, 'enc_tail_dgrad_x6_kernel')
        entries['tvae_enc_tail_wgrad_x6'] = (2.0 * c['C'] * c['C'] * ncol, 'enc_tail_wgrad_x6_kernel')
        tail_bytes = {'tvae_enc_tail_fwd_x6': (2 * 4 * c['C'] + 4 * nh_ + 32) * ncol,       # A1 in, H out, heads, sign words
                      'tvae_enc_tail_dgrad_x6': (4 * c['C'] + 4 * nh_ + 32) * ncol,         # dA1 out, head gradients, sign words
                      'tvae_enc_tail_wgrad_x6': (4 * c['C'] + 4 * nh_ + 16) * ncol}         # A1 in, head gradients, sign words of H
        timed = {k_: v for k_, v in kev.items() if k_ in entries}
        enc_tail = {k_: {'ms': round(kev[k_]['mean_ms'], 3), 'bound': 'hbm',
                         'algorithmic_bytes_per_launch': float(tail_bytes[k_]),
                         'achieved_GBps': tail_bytes[k_] / (kev[k_]['mean_ms'] * 1e-3) / 1e9,
                         'frac_of_8TBps': tail_bytes[k_] / (kev[k_]['mean_ms'] * 1e-3) / 8.0e12,
                         'mfma_per_block_as_launched': kev[k_].get('mfma_per_block_seen'),
                         'executed_bf16_pflops': (kev[k_].get('mfma_per_block') or {'bf16': 1, 'h3': 3}.get(mode, 6)) *
                         entries[k_][0] / (kev[k_]['mean_ms'] * 1e-3) / 1e15}
                    for k_ in tail_bytes if k_ in kev}
        # the roofline object describes the dominant KERNEL FAMILY of the step, the split-pipe dense GEMM (decoder layers
        # and the spectral contraction of the convolution), on its LARGEST decoder launch (not its best one), with the
        # call-weighted aggregate over the three decoder launches beside it.  bf16 MFMAs per product block: 6 in the
        # exact-split arithmetic, 3 where the streamed operand is the exact 0 / 1 matrix [H > 0] (two-valued implicit
        # LeakyReLU gradient: data / weight gradient of the last hidden layer).  In fp32 mode the dominant kernel is the
        # direct lifting-convolution weight gradient.
        two_val = c['layers'] >= 2 and c['n_out'] == 1          # ops.DecoderFn: virt + LeakyReLU (bench models)
        products = {'tvae_linear_fwd_x6': 6, 'tvae_linear_dgrad_x6': 3 if two_val else 6,
                    'tvae_linear_wgrad_x6': 3 if two_val else 6}
        if mode == 'h3':        # two fp16 parts: 3 products, 2 against the exact 0 / 1 operand (the general forms stay x6)
            products = {'tvae_linear_fwd_x6': 3, 'tvae_linear_dgrad_x6': 2 if two_val else 6,
                        'tvae_linear_wgrad_x6': 2 if two_val else 6}
        if mode == 'bf16':
            products = {k_: 1 for k_ in products}
        # ... as the MODE suggests; what counts is what was LAUNCHED: tvae.ops records the `parts` it passed to every entry
        # point and the matrix instructions per product block that follow from them (ops._timed / mfma_per_block); a launch
        # that fell back to the three-part split is then priced as six products, not three
        products_by_mode = dict(products)
        products = {k_: (kev.get(k_, {}).get('mfma_per_block') or products[k_]) for k_ in products}
        dense = [k_ for k_ in products if k_ in timed]
        if mode in ('x6', 'h3', 'bf16') and dense:
            dom = max(dense, key=lambda k_: kev[k_]['total_ms'])
        else:
            dom = max(('tvae_conv1_fwd', 'tvae_conv1_wgrad'), key=lambda k_: kev.get(k_, {}).get('total_ms', 0.0))
        flops = entries[dom][0]
        ach = flops / (kev[dom]['mean_ms'] * 1e-3) / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS / products[dom] if (mode in ('x6', 'h3', 'bf16') and dom in products) else info['peak']
        out = {
            'metric': ('training images/sec (fwd+bwd+Adam), P8 z=2 64x64 bs=256/GPU' if args.workload == 'S64' else
                       'training images/sec (fwd+bwd+Adam), extra workload ' + args.workload) +
                      (' [bf16 throughput mode: extra measurement, not the headline]' if mode == 'bf16' else ''),
            'value': imgs / dt, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': info['dtype'], 'data': 'synthetic',
            'elbo': elbo_last,
            'config': {'workload': c['desc'],
                       'global_batch': world * B, 'per_gpu_batch': B, 'parallelism': f'dp{world}',
                       'collective': ({'backend': dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else ' (REHEARSAL on one GPU: '
                                                                            'numbers are not a scaling measurement)'),
                                       'world_size_seen': dist.get_world_size(),
                                       'payload_bytes_per_step': int(opt.flat_g.numel()) * 4,
                                       'early_bucket_bytes': int(opt._early_end) * 4,
                                       'diagnostics': dp_diag,
                                       'buckets': 'decoder segment posted from the backward, encoder segment at the '
                                                  'optimizer step (tvae/optim.py)'}
                                      if world > 1 else None),
                       'arithmetic_mode': mode, 'graph_replay': bool(args.graph),
                       'lifting_conv': (('mixed domain: DFT along x on the short circular frame L = %d (n + pad, not n + 2 pad), tap '
                                         'rows spatial; batched split-pipe GEMM with reduction 2*k*Cin = %d: %.0f GFLOP of matrix work '
                                         'per launch instead of %.0f in the direct form') %
                                        (_lib.query('tvae_conv1_dft_frame', B, c['cin'], c['n'], c['k'], c['pad'], c['C'], c['R']),
                                         2 * c['k'] * c['cin'],
                                         2.0 * (_lib.query('tvae_conv1_dft_frame', B, c['cin'], c['n'], c['k'], c['pad'], c['C'], c['R']) // 2 + 1)
                                         * 2 * c['C'] * c['R'] * 2 * c['k'] * c['cin'] * B * (c['n'] + 2 * c['pad'] - c['k'] + 1) / 1e9,
                                         conv_flops / 1e9)
                                        if conv_dft else 'direct implicit GEMM')},
            'roofline': {'kernel': dom + ' (' + entries[dom][1] + ', ' + info['insn'] + ')', 'bound': 'mfma',
                         'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': ach / peak,
                         'peak_note': 'algorithmic (fp32-equivalent) FLOP/s; peak = 2500 TFLOP/s dense bf16 / bf16 MFMAs per '
                                      'product block of this launch (dense_launches)' if mode in ('x6', 'h3', 'bf16') else
                                      'dense f32 MFMA peak',
                         'frac_of_f32_mfma_peak': ach / PEAK_F32_MFMA_TFLOPS,
                         # dense_x6_kernel is launched with several shapes: pick the decoder-layer launch by its grid
                         # (512 threads x 8*ceil(tiles_n/8) workgroups; forward and data-gradient launches averaged)
                         'traffic': pmc_traffic(dom, 512 * 8 * ((Nt // 128 + 7) // 8) * ((c['hidden'] + 511) // 512)
                                                if dom.startswith('tvae_linear') else None,
                                                kernel=entries[dom][1].split()[0] if dom.startswith('tvae_linear') else None)
                         if args.workload == 'S64' else None,
                         'algorithmic_flops_per_launch': flops, 'mean_launch_ms': kev[dom]['mean_ms'],
                         'launches_timed': kev[dom]['launches'],
                         'entry_points_ms': {k_: round(v['mean_ms'], 3) for k_, v in sorted(timed.items())},
                         'dense_launches': {k_: {'ms_per_step': round(kev[k_]['total_ms'] / args.steps, 3),
                                                 'launches_per_step': kev[k_]['launches'] / args.steps,
                                                 'bf16_mfma_per_product_block': products[k_],
                                                 'mfma_per_block_as_launched': kev[k_].get('mfma_per_block_seen'),
                                                 'mfma_per_block_expected_for_mode': products_by_mode[k_],
                                                 'algorithmic_tflops': dense_flops * kev[k_]['launches'] /
                                                 (kev[k_]['total_ms'] * 1e-3) / 1e12,
                                                 'executed_bf16_pflops': products[k_] * dense_flops * kev[k_]['launches'] /
                                                 (kev[k_]['total_ms'] * 1e-3) / 1e15,
                                                 'frac_of_bf16_peak': products[k_] * dense_flops * kev[k_]['launches'] /
                                                 (kev[k_]['total_ms'] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS}
                                            for k_ in dense} if mode in ('x6', 'h3', 'bf16') else None,
                         'dense_aggregate': ({'algorithmic_tflops': sum(dense_flops * kev[k_]['launches'] for k_ in dense) /
                                              (sum(kev[k_]['total_ms'] for k_ in dense) * 1e-3) / 1e12,
                                              'executed_bf16_pflops':
                                              sum(products[k_] * dense_flops * kev[k_]['launches'] for k_ in dense) /
                                              (sum(kev[k_]['total_ms'] for k_ in dense) * 1e-3) / 1e15,
                                              'frac_of_bf16_peak':
                                              sum(products[k_] * dense_flops * kev[k_]['launches'] for k_ in dense) /
                                              (sum(kev[k_]['total_ms'] for k_ in dense) * 1e-3) / 1e12 /
                                              PEAK_BF16_MFMA_TFLOPS}
                                             if (mode in ('x6', 'h3', 'bf16') and dense) else None),
                         'conv_direct_form_tflops': {k_: conv_flops / (kev[k_]['mean_ms'] * 1e-3) / 1e12
                                                     for k_ in ('tvae_conv1_fwd', 'tvae_conv1_wgrad') if k_ in kev}},
        }
        if strong is not None:
            out['strong_scaling'] = strong
        if small_batch is not None:
            out['small_batch'] = small_batch
        if workloads is not None:
            out['workloads'] = workloads
        if enc_fwd is not None:
            out['encoder_forward'] = enc_fwd
        if enc_inf is not None:
            out['encoder_forward_inference'] = enc_inf
        if enc_tail:
            out['encoder_tail'] = enc_tail
        if companion is not None:
            out['exact_f32_mode'] = companion
        if companion_x6 is not None:
            out['exact_split_x6_mode'] = companion_x6
        if companion_bf16 is not None:
            out['bf16_throughput_mode'] = companion_bf16
        if companion_graph is not None:
            out['graph_replay_mode'] = companion_graph
        if world == 1 and not args.no_cpu_baseline and args.workload == 'S64':
            out['cpu_baseline'] = cpu_baseline()
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        if reducer is not None:
            reducer.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
