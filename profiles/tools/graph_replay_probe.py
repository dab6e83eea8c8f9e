#!/usr/bin/env python3
"""hipGraph replay of the forward + backward of one training step against eager execution, BIT FOR BIT, with the two
triggers that corrupted replays in round 2 (profiles/experiments/README.md): a host synchronize between a replay and the
next launch, and `copy.deepcopy(generator).cpu()` between replays.

    python profiles/tools/graph_replay_probe.py [S28|S28F|S64s] [replays]

Protocol: static inputs (minibatch, noise), parameters fixed; eager reference gradients g_ref; capture on a side stream
after a warm-up on it (every workspace filled with NaN first, so an accumulate-into-unwritten-memory would show); then
`replays` rounds of {replay, trigger, compare flat_g and the ELBO terms with the eager reference bitwise}."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import numpy as np
import torch

CFG = {'S28': (28, 8, 256, 128, 512, 28, 8, False, 'bce'), 'S28F': (28, 16, 64, 128, 512, 28, 8, True, 'bce'),
       'S64s': (64, 8, 32, 128, 512, 64, 16, False, 'gauss')}


def main():
    import src.models as M
    from tvae import ops, optim, step, tables
    name = sys.argv[1] if len(sys.argv) > 1 else 'S28'
    replays = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    n, R, B, C, hid, k, pad, four, lik = CFG[name]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    gen = M.SpatialGenerator(2, hid, num_layers=2, fourier_expansion=four, sigma=2.0 / (n - 1)).to(dev)
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, 1, 2, kernels_num=C, kernels_size=k, padding=pad, groupconv=R, rot_refinement=True, theta_prior=np.pi,
        normal_prior_over_r=False).to(dev)
    params = list(gen.parameters()) + list(enc.parameters())
    opt = optim.FlatAdam(params, lr=2e-4)
    x = torch.from_numpy(tables.image_coords(n)).to(dev)
    step.pixel_spacing(x)
    y = torch.rand(B, 1, n, n, device=dev)
    ho = n + 2 * pad - k + 1
    noise = step.draw_noise(B, R * ho * ho, 2, dev)

    def fwd_bwd():
        elbo, lp, kl = step.elbo_terms(x, y, gen, enc, lik, noise)
        (-elbo).backward()
        return torch.stack([elbo.detach().double(), lp.detach().double(), kl.detach().double()])

    # eager repeatability first: steps 1, 2, 3 of this process on fixed inputs (step 1 allocates every scratch buffer)
    runs = []
    for i in range(3):
        opt.zero_grad()
        t_ = fwd_bwd().clone()
        torch.cuda.synchronize()
        runs.append((t_, opt.flat_g.clone()))
    for i in (0, 1):
        same = torch.equal(runs[i][1].view(torch.int32), runs[2][1].view(torch.int32)) and torch.equal(runs[i][0], runs[2][0])
        print(f'eager step {i + 1} vs step 3: {"bitwise equal" if same else "DIFFERENT"}', flush=True)
        if not same:
            off = 0
            for nm, p in [('d.' + k_, v) for k_, v in gen.named_parameters()] + [('e.' + k_, v) for k_, v in enc.named_parameters()]:
                a = runs[i][1][opt._offsets[off]:opt._offsets[off] + p.numel()]
                b = runs[2][1][opt._offsets[off]:opt._offsets[off] + p.numel()]
                off += 1
                if not torch.equal(a.view(torch.int32), b.view(torch.int32)):
                    print('      ', nm, int((a != b).sum()), 'of', a.numel(), 'elements, max rel', float((a - b).abs().max() / b.abs().max()))
    ref_terms, g_ref = runs[2]
    # warm-up on the capture stream, workspaces poisoned, then capture
    for t in ops._WS.values():
        t.fill_(float('nan'))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            opt.zero_grad()
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if not torch.equal(opt.flat_g.view(torch.int32), g_ref.view(torch.int32)):
        # a workspace is read before it is written: find out which one (poison them one at a time, default stream)
        print('eager run after poisoning the workspaces differs from the reference; bisecting over the scratch buffers')
        off = 0
        for nm, p in [('d.' + k_, v) for k_, v in gen.named_parameters()] + [('e.' + k_, v) for k_, v in enc.named_parameters()]:
            gv, gr = p.grad.reshape(-1), g_ref[opt._offsets[off]:opt._offsets[off] + p.numel()]
            off += 1
            if not torch.equal(gv.view(torch.int32), gr.view(torch.int32)):
                print('   differs:', nm, 'nan' if torch.isnan(gv).any() else float((gv - gr).abs().max() / gr.abs().max()))
        for key in list(ops._WS):
            for t in ops._WS.values():
                t.zero_()
            ops._WS[key].fill_(float('nan'))
            opt.zero_grad()
            fwd_bwd()
            torch.cuda.synchronize()
            same = torch.equal(opt.flat_g.view(torch.int32), g_ref.view(torch.int32))
            print('   poison', key[0], ops._WS[key].numel(), 'floats ->', 'same' if same else 'DIFFERENT', flush=True)
        return 2
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad()
    with torch.cuda.graph(graph, stream=side):
        terms = fwd_bwd()
    torch.cuda.synchronize()
    results = {}
    for trig in ('none', 'sync', 'deepcopy_cpu', 'zero_then_sync'):
        bad = []
        for i in range(replays):
            opt.flat_g.zero_()
            if trig == 'zero_then_sync':
                torch.cuda.synchronize()
            graph.replay()
            if trig == 'sync':
                torch.cuda.synchronize()
            elif trig == 'deepcopy_cpu':
                copy.deepcopy(gen).cpu()
            g = opt.flat_g.clone()                       # (the "next launch" after the replay)
            t = terms.clone()
            torch.cuda.synchronize()
            if not torch.equal(g.view(torch.int32), g_ref.view(torch.int32)) or not torch.equal(t, ref_terms):
                d = (g.double() - g_ref.double()).abs()
                bad.append((i, int((g.view(torch.int32) != g_ref.view(torch.int32)).sum()), float(d.nan_to_num(1e30).max()),
                            [float(v) for v in (t - ref_terms).abs()]))
        results[trig] = bad
        print(f'{name} trigger {trig:14s}: {len(bad)} of {replays} replays differ from eager', bad[:3], flush=True)
    return 0 if not any(results.values()) else 1


if __name__ == '__main__':
    sys.exit(main())
