"""Data-parallel path on CPU: 2 processes over gloo run the product's sharding / gradient all-reduce / flat-Adam
logic (tvae.dp, tvae.optim) and must reproduce a single-process run on the same global minibatches.
Per-rank gradients come from the CPU oracle (the HIP kernels need a GPU); on the GPU box the same wrapper runs
over RCCL (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, rel_err

CFG = dict(R=4, padding=2, rot_refinement=True, theta_prior=np.pi, normal_prior_over_r=False, num_layers=2)
N_IMG, GB, NPIX, ZD, HO = 11, 6, 12, 2, 5


def _make_params():
    import src.models as M
    torch.manual_seed(3)
    gen = M.SpatialGenerator(ZD, 8, num_layers=2)
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        NPIX, 1, ZD, kernels_num=4, kernels_size=12, padding=2, groupconv=4, rot_refinement=True,
        theta_prior=np.pi, normal_prior_over_r=False)
    with torch.no_grad():
        for m in (enc.conv_a, enc.conv_r, enc.conv_z):
            m.weight.mul_(10.0)
    return gen, enc


def _data_and_noise(n_img=N_IMG):
    g = torch.Generator().manual_seed(11)
    data = torch.rand(n_img, 1, NPIX, NPIX, generator=g)
    E = torch.empty(n_img, 4 * HO * HO).exponential_(generator=g)
    return data, E, torch.randn(n_img, ZD, generator=g), torch.randn(n_img, generator=g)


def _torch_adam(p, g, m, v, step, lr, b1, b2, eps, scale):
    from oracle import tvae_oracle as O
    O.adam_update([p], [g * scale], [m], [v], step, lr, b1, b2, eps)


def _train(rank, world, epochs=2, N_IMG=N_IMG, GB=GB, graph_from=None):
    """graph_from = e: from epoch e on the step follows tvae/graph.py's protocol -- `disable_early_bucket()` once (a captured
    backward must not post collectives), full-size shards accumulate STRAIGHT into the flat gradient buffer (zeroed, every
    p.grad its view: what a replay leaves behind), any other shard size runs the eager set_to_none protocol."""
    from oracle import tvae_oracle as O
    from tvae import dp, optim
    gen, enc = _make_params()
    params = list(gen.parameters()) + list(enc.parameters())
    reducer = dp.GradReducer() if world > 1 else None
    opt = optim.FlatAdam(params, lr=1e-2, reducer=reducer, update_fn=_torch_adam,
                          early_params=len(list(gen.parameters())))
    data, E, ez, et = _data_and_noise(N_IMG)
    x = O.image_coords(NPIX)
    batches = dp.ShardedBatches(data, GB, rank, world, shuffle=True, seed=5, reducer=reducer)
    stats = [0.0, 0.0]
    full = -(-GB // world)                               # the shard size a graph would have been captured at
    n_early_eager = 0
    for ep in range(epochs):
        batches.set_epoch(ep)
        perm = dp.epoch_permutation(N_IMG, 5, ep)
        graph_mode = graph_from is not None and ep >= graph_from
        if graph_mode and ep == graph_from:
            opt.disable_early_bucket()                   # GraphedStep.__init__
        for (y,), (lo, hi, g) in zip(batches, dp.shard_slices(N_IMG, GB, rank, world)):
            idx = perm[lo:hi]
            assert torch.equal(y, data[idx])
            if not graph_mode:
                n_early_eager += 1
            if hi == lo:        # no image of this global minibatch on this rank: still in both collectives, weight 0 (tvae/step.py)
                opt.step()
                opt.zero_grad(set_to_none=True)
                continue
            if graph_mode and hi - lo == full:           # GraphedStep._fwd_bwd / run: gradients land in the flat buffer
                opt.flat_g.zero_()
                for p_, gv in zip(opt._ps, opt._gviews):
                    p_.grad = gv
            encp = dict(enc.named_parameters())
            genp = dict(gen.named_parameters())
            elbo, _, _ = O.elbo_step(x, y, encp, genp, likelihood='bce', E=E[idx], eps_z=ez[idx],
                                     eps_theta=et[idx], **CFG)
            (-elbo).backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            stats[0] += float(elbo) * (hi - lo)
            stats[1] += hi - lo
    if world > 1:           # two buckets: the decoder segment was posted from inside every EAGER-phase backward (tvae/optim.py)
        assert reducer.posted_early == n_early_eager and not reducer._pending
    stats = dp.allreduce_stats(stats, torch.device('cpu'))
    named = {'d.' + k_: v.detach().clone() for k_, v in gen.named_parameters()}
    named.update({'e.' + k_: v.detach().clone() for k_, v in enc.named_parameters()})
    return named, stats


def _worker(rank, world, port, out_dir, kw=None):
    for p in (ROOT, os.path.join(ROOT, 'target-vae_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from tvae import dp
    r, w, _ = dp.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    flat, stats = _train(rank, world, **(kw or {}))
    torch.save(dict(flat=flat, stats=stats), os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(600)
def test_two_rank_gloo_matches_single_process(tmp_path):
    torch.set_num_threads(4)
    flat1, stats1 = _train(0, 1)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    for k_ in flat1:
        assert torch.equal(r0['flat'][k_], r1['flat'][k_]), k_        # replicas stay bit-identical
        if k_ == 'e.conv_a.bias':
            continue      # analytic gradient 0 (softmax shift invariance): Adam follows rounding noise
        assert rel_err(r0['flat'][k_], flat1[k_]) < 2e-4, k_          # == the single-process global-batch run
    assert r0['stats'][1] == stats1[1] == 2 * N_IMG
    assert abs(r0['stats'][0] - stats1[0]) / abs(stats1[0]) < 1e-5


@pytest.mark.timeout(900)
def test_eight_rank_gloo_reference_minibatch_matches_single_process(tmp_path):
    """EIGHT ranks (the node the north star names) over gloo with the reference's default minibatch of
    100 images (train_mnist.py:426): every global minibatch splits 13 / 12 over the ranks, the ragged tail (205 = 2 x 100 + 5
    images) leaves three ranks with NO image -- they still join both buckets of the all-reduce with weight 0 and take the
    same Adam step -- and all eight replicas must stay bit-identical and equal the single process on the global batches."""
    kw = dict(epochs=1, N_IMG=205, GB=100)
    torch.set_num_threads(4)
    flat1, stats1 = _train(0, 1, **kw)
    from tvae import dp
    sizes = [[hi - lo for lo, hi, _ in dp.shard_slices(205, 100, r, 8)] for r in range(8)]
    assert sorted(s_[0] for s_ in sizes) == [12] * 4 + [13] * 4 and [s_[2] for s_ in sizes].count(0) == 3
    mp.spawn(_worker, args=(8, _free_port(), str(tmp_path), kw), nprocs=8, join=True)
    rs = [torch.load(tmp_path / f'rank{r}.pt') for r in range(8)]
    for k_ in flat1:
        for r in range(1, 8):
            assert torch.equal(rs[0]['flat'][k_], rs[r]['flat'][k_]), (k_, r)
        if k_ == 'e.conv_a.bias':
            continue
        assert rel_err(rs[0]['flat'][k_], flat1[k_]) < 2e-4, k_
    assert rs[0]['stats'][1] == stats1[1] == 205
    assert abs(rs[0]['stats'][0] - stats1[0]) / abs(stats1[0]) < 1e-5


# ----------------------------------------------------------------------------------------------------------------
# shard-RESIDENT loop (what the drivers use): a rank holds only its contiguous slice of the dataset; the global
# minibatches are a shuffle stratified by shard.  Two gloo ranks == one process walking dp.resident_global_batches.
def _train_resident(rank, world, emulate=2, epochs=2, GB=GB):
    from oracle import tvae_oracle as O
    from tvae import dp, optim
    gen, enc = _make_params()
    params = list(gen.parameters()) + list(enc.parameters())
    reducer = dp.GradReducer() if world > 1 else None
    opt = optim.FlatAdam(params, lr=1e-2, reducer=reducer, update_fn=_torch_adam,
                          early_params=len(list(gen.parameters())))
    data, E, ez, et = _data_and_noise()
    x = O.image_coords(NPIX)
    n_img = N_IMG - 3                                   # 8 images, global minibatches of 6: ragged tail, quota 3 per rank
    data, E, ez, et = data[:n_img], E[:n_img], ez[:n_img], et[:n_img]
    r0, r1 = dp.shard_bounds(n_img, rank, world)
    it = dp.ResidentShardBatches(data[r0:r1].clone(), n_img, GB, rank, world, shuffle=True, seed=5, reducer=reducer)
    stats = [0.0, 0.0]
    for ep in range(epochs):
        it.set_epoch(ep)
        glob = dp.resident_global_batches(n_img, GB, emulate, 5, ep)
        if world > 1:                                    # this rank's part of every global minibatch, as row indices
            lperm = dp.local_permutation(r1 - r0, 5, ep, rank) + r0
            plan, pos, mine = dp.shard_plan(n_img, GB, world), 0, []
            for counts in plan:
                mine.append(lperm[pos:pos + counts[rank]])
                pos += counts[rank]
            for gidx, counts, m_ in zip(glob, plan, mine):      # the ranks' parts tile the global minibatch
                off = sum(counts[:rank])
                assert torch.equal(gidx[off:off + counts[rank]], m_)
        else:
            mine = glob
        for (y,), idx in zip(it if world > 1 else [(data[i],) for i in glob], mine):
            assert torch.equal(y, data[idx])
            if world > 1:
                reducer_ok = True                        # fraction set by the iterator
            if len(idx) > 0:
                elbo, _, _ = O.elbo_step(x, y, dict(enc.named_parameters()), dict(gen.named_parameters()),
                                         likelihood='bce', E=E[idx], eps_z=ez[idx], eps_theta=et[idx], **CFG)
                (-elbo).backward()
                stats[0] += float(elbo) * len(idx)
                stats[1] += len(idx)
            opt.step()
            opt.zero_grad()
    stats = dp.allreduce_stats(stats, torch.device('cpu'))
    named = {'d.' + k_: v.detach().clone() for k_, v in gen.named_parameters()}
    named.update({'e.' + k_: v.detach().clone() for k_, v in enc.named_parameters()})
    return named, stats


def _worker_resident(rank, world, port, out_dir, gb=GB):
    for p in (ROOT, os.path.join(ROOT, 'target-vae_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from tvae import dp
    dp.init_from_env(backend='gloo')
    flat, stats = _train_resident(rank, world, GB=gb)
    torch.save(dict(flat=flat, stats=stats), os.path.join(out_dir, f'res_rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('gb', [6, 5])       # 5: global minibatch not divisible by the ranks (3+2, then 1+2 of 8 images)
def test_two_rank_resident_shards_match_single_process(tmp_path, gb):
    torch.set_num_threads(4)
    from tvae import dp
    assert [sum(c) for c in dp.shard_plan(N_IMG - 3, gb, 2)] == [gb, N_IMG - 3 - gb]
    flat1, stats1 = _train_resident(0, 1, GB=gb)
    mp.spawn(_worker_resident, args=(2, _free_port(), str(tmp_path), gb), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'res_rank0.pt')
    r1 = torch.load(tmp_path / 'res_rank1.pt')
    for k_ in flat1:
        assert torch.equal(r0['flat'][k_], r1['flat'][k_]), k_
        if k_ == 'e.conv_a.bias':
            continue
        assert rel_err(r0['flat'][k_], flat1[k_]) < 2e-4, k_
    assert r0['stats'][1] == stats1[1] == 2 * (N_IMG - 3)
    assert abs(r0['stats'][0] - stats1[0]) / abs(stats1[0]) < 1e-5


# ----------------------------------------------------------------------------------------------------------------
# ragged tail smaller than the number of ranks (empty shard) + random Fourier BUFFERS that differ per rank at
# construction: the product's train_epoch / eval_model / broadcast_buffers must keep the replicas identical and equal to
# the single-process run
N2, GB2 = 13, 6          # global minibatches of 6, 6, 1 -> rank 1 holds no image of the last one


def _train_epoch_product(rank, world):
    from oracle import tvae_oracle as O
    from tvae import dp, optim, step
    import src.models as M
    torch.manual_seed(3 + 17 * rank)            # replicas start DIFFERENT (unseeded construction in the driver)
    gen = M.SpatialGenerator(ZD, 8, num_layers=2, fourier_expansion=True, sigma=2.0 / (NPIX - 1))
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        NPIX, 1, ZD, kernels_num=4, kernels_size=12, padding=2, groupconv=4, rot_refinement=True,
        theta_prior=np.pi, normal_prior_over_r=False)
    params = list(gen.parameters()) + list(enc.parameters())
    reducer = dp.GradReducer() if world > 1 else None
    opt = optim.FlatAdam(params, lr=1e-2, reducer=reducer, update_fn=_torch_adam,
                          early_params=len(list(gen.parameters())))
    if world > 1:
        dist.broadcast(opt.flat_p, src=0)
        dp.broadcast_buffers(gen, enc)
    g = torch.Generator().manual_seed(11)
    data = torch.rand(N2, 1, NPIX, NPIX, generator=g)
    E = torch.empty(N2, 4 * HO * HO).exponential_(generator=g)
    ez, et = torch.randn(N2, ZD, generator=g), torch.randn(N2, generator=g)
    x = O.image_coords(NPIX)
    cfg = dict(CFG, fourier_sigma=2.0 / (NPIX - 1))

    def oracle_minibatch(xc, y, gm, em, t_inf, r_inf, epoch, device, theta_prior, groupconv, image_dim, likelihood='bce',
                         noise=None):
        gsd = dict(gm.named_parameters())
        gsd.update(dict(gm.named_buffers()))
        return O.elbo_step(xc, y, dict(em.named_parameters()), gsd, likelihood='bce', E=noise[0], eps_z=noise[1],
                           eps_theta=noise[2], **cfg)

    step.eval_minibatch = oracle_minibatch       # the HIP kernels need a GPU; the host logic under test is the loop
    batches = dp.ShardedBatches(data, GB2, rank, world, shuffle=True, seed=5, reducer=reducer)
    perm = dp.epoch_permutation(N2, 5, 0)
    noise = [(E[perm[lo:hi]], ez[perm[lo:hi]], et[perm[lo:hi]]) for lo, hi, _ in dp.shard_slices(N2, GB2, rank, world)]
    sizes = [hi - lo for lo, hi, _ in dp.shard_slices(N2, GB2, rank, world)]
    tr = step.train_epoch(batches, x, gen, enc, opt, 'attention', 'attention+offsets', 0, 1, max(sum(sizes), 1), 'cpu',
                          params, np.pi, 4, NPIX, progress=False, noise_iter=iter(noise))
    test_it = dp.ShardedBatches(data, GB2, rank, world, shuffle=False, seed=5)
    noise_t = [(E[lo:hi], ez[lo:hi], et[lo:hi]) for lo, hi, _ in dp.shard_slices(N2, GB2, rank, world)]
    ev = step.eval_model(test_it, x, gen, enc, 'attention', 'attention+offsets', 0, 'cpu', np.pi, 4, NPIX,
                         noise_iter=iter(noise_t))
    tot = dp.allreduce_stats([tr[0] * sum(sizes), ev[0] * sum(sizes), float(sum(sizes))], torch.device('cpu'))
    named = {'d.' + k_: v.detach().clone() for k_, v in gen.state_dict().items()}
    named.update({'e.' + k_: v.detach().clone() for k_, v in enc.state_dict().items()})
    return named, tot, sizes


def _worker2(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, 'target-vae_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from tvae import dp
    dp.init_from_env(backend='gloo')
    named, tot, sizes = _train_epoch_product(rank, world)
    torch.save(dict(named=named, tot=tot, sizes=sizes), os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_empty_shard_and_fourier_buffers(tmp_path):
    torch.set_num_threads(4)
    from tvae import step
    keep = step.eval_minibatch
    try:
        one, tot1, _ = _train_epoch_product(0, 1)
    finally:
        step.eval_minibatch = keep
    mp.spawn(_worker2, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    assert r0['sizes'] == [3, 3, 1] and r1['sizes'] == [3, 3, 0]                 # rank 1's last shard is empty
    for k_ in one:
        assert torch.isfinite(r1['named'][k_]).all(), k_                          # no NaN from an empty mean
        assert torch.equal(r0['named'][k_], r1['named'][k_]), k_                  # replicas identical, buffers included
        if k_ == 'e.conv_a.bias':
            continue
        assert rel_err(r0['named'][k_], one[k_]) < 2e-4, k_                       # == single process, global batches
    assert r0['tot'][2] == tot1[2] == N2
    for i in (0, 1):
        assert abs(r0['tot'][i] - tot1[i]) / abs(tot1[i]) < 1e-5


@pytest.mark.timeout(600)
def test_two_rank_gloo_graph_protocol_after_early_bucket(tmp_path):
    """VERDICT r05 item 8 / ADVICE r04: the two-bucket all-reduce and the hipGraph step together.  Epoch 0 runs eagerly with the
    early decoder bucket posted from inside every backward; then the optimizer is switched the way GraphedStep does
    (`disable_early_bucket()`: hooks stay registered, `_early_n = 0`) and full-size shards accumulate straight into the flat
    buffer while ragged ones keep the eager set_to_none protocol.  Both ranks must stay bit-identical and follow the
    single-process run of the same schedule; no early bucket may be posted after the switch."""
    torch.set_num_threads(4)
    kw = dict(epochs=3, graph_from=1)
    flat1, stats1 = _train(0, 1, **kw)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), kw), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    for k_ in flat1:
        assert torch.equal(r0['flat'][k_], r1['flat'][k_]), k_
        if k_ == 'e.conv_a.bias':
            continue
        assert rel_err(r0['flat'][k_], flat1[k_]) < 3e-4, k_
    assert r0['stats'][1] == stats1[1] == 3 * N_IMG
