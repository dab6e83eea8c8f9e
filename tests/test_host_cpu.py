"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol of include/tvae_hip.h,
host tables equal the oracle's, the drop-in classes reproduce the reference's parameter names / shapes /
default initialisation, checkpoints pickle as `src.models.*`, and the product path refuses CPU tensors."""
import io
import os
import pickle
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, rel_err, tdict
from oracle import tvae_oracle as O


def test_library_exports_every_declared_symbol():
    from tvae import _lib
    hdr = open(os.path.join(ROOT, 'include', 'tvae_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(?:int|long)\s+(tvae_\w+)\s*\(', hdr)))
    assert declared == sorted(_lib.exported_symbols())
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.tvae_abi_version() == _lib.ABI_VERSION == 7


def test_signature_arity_matches_header():
    from tvae import _lib
    hdr = open(os.path.join(ROOT, 'include', 'tvae_hip.h')).read()
    for name, sig in _lib.SIGNATURES.items():
        m = re.search(r'\bint\s+' + name + r'\s*\(([^;]*?)\)\s*;', hdr, re.S)
        assert m, name
        args = [a.strip() for a in m.group(1).split(',')]
        assert args[-1].startswith('tvae_stream_t'), name
        assert len(args) - 1 == len(sig), (name, len(args) - 1, len(sig))
        for a, c in zip(args[:-1], sig):
            if c == 'p':
                assert '*' in a, (name, a)
            elif c == 'f':
                assert a.startswith('float ') and '*' not in a, (name, a)
            elif c == 'l':
                assert a.startswith('long '), (name, a)
            else:
                assert a.startswith('int ') and '*' not in a, (name, a)


def test_product_path_refuses_cpu_tensors():
    import src.models as M
    from tvae import _lib
    gc = M.GroupConv(1, 2, 5, output_rot_dim=4)
    with pytest.raises(RuntimeError):
        gc(torch.rand(1, 1, 8, 8), 'cpu')
    with pytest.raises(_lib.TvaeHipError):
        _lib.call('tvae_act_bwd', torch.zeros(4), torch.zeros(4), torch.zeros(4), 4, 1, 0.01)
    gen = M.SpatialGenerator(2, 8)
    with pytest.raises(RuntimeError):
        gen(torch.zeros(1, 4, 2), torch.zeros(1, 2))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'target-vae_amd')
    for d, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(d, f)).read()
                assert 'oracle' not in src.replace('no CPU fallback; use oracle', ''), os.path.join(d, f)


@pytest.mark.parametrize('k,R', [(5, 4), (28, 8), (9, 16), (64, 8)])
def test_tap_tables_match_oracle(k, R):
    from tvae import tables
    i1, w1 = tables.rotation_taps(k, R)
    i2, w2 = O.rotation_taps(k, R)
    assert np.array_equal(i1, i2.astype(np.int32)) and np.array_equal(w1, w2)
    if k > 28:
        return                                   # dense check below is O(R k^4) memory
    ptr, er, ed, ew = tables.rotation_taps_csr(k, R)
    dense = np.zeros((R, k * k, k * k), dtype=np.float64)          # [r][dst][src]
    for t in range(4):
        ok = i1[:, :, t] >= 0
        rr, dd = np.nonzero(ok)
        np.add.at(dense, (rr, dd, i1[rr, dd, t]), w1[rr, dd, t])
    dense2 = np.zeros_like(dense)
    for s in range(k * k):
        for e in range(ptr[s], ptr[s + 1]):
            dense2[er[e], ed[e], s] += ew[e]
    assert np.allclose(dense, dense2, atol=1e-7)


@pytest.mark.parametrize('R,refine,normal,tp', [(8, True, False, np.pi), (16, True, True, np.pi / 4),
                                                (4, False, False, np.pi)])
def test_prior_tables_match_oracle(R, refine, normal, tp):
    from tvae import tables
    assert np.allclose(tables.rotation_log_prior(R, refine, tp, normal), O.rotation_log_prior(R, refine, tp, normal),
                       rtol=1e-6, atol=1e-6)
    off = O.rotation_offsets(R) if refine else np.zeros(R, np.float32)
    assert np.array_equal(tables.rotation_offsets(R, refine), off)
    for Ho in (17, 33, 8):
        s = 2.0 / 27
        assert np.array_equal(tables.translation_grid(Ho, s), O.translation_grid(Ho, s))
        G = torch.from_numpy(O.translation_grid(Ho, s))
        p_t = torch.distributions.Normal(torch.tensor([0.0]), torch.tensor([0.1])).log_prob(G).sum(1)
        p_r = torch.from_numpy(O.rotation_log_prior(R, refine, tp, normal))
        ref = torch.log_softmax((p_t.view(1, -1) + p_r.view(R, 1)).reshape(-1), 0)
        got = tables.joint_log_prior(Ho, s, tables.rotation_log_prior(R, refine, tp, normal))
        assert np.allclose(got, ref.numpy(), rtol=1e-5, atol=1e-4)
    assert np.array_equal(tables.image_coords(28), O.image_coords(28).numpy())


def test_default_init_and_names_match_reference():
    """Same seed + same construction order (generator first, train_mnist.py:522,551) => identical parameters to
    the reference-generated fixture; state_dict names / shapes are the reference's."""
    import src.models as M
    fx = load_golden('step_mnist28_P8_init')
    torch.manual_seed(0)
    gen = M.SpatialGenerator(2, 512, n_out=1, num_layers=2, resid=False, fourier_expansion=False, sigma=2.0 / 27)
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        28, 1, 2, kernels_num=128, kernels_size=28, padding=8, groupconv=8, rot_refinement=True, theta_prior=np.pi,
        normal_prior_over_r=False)
    e_ref, d_ref = tdict(fx, 'e.'), tdict(fx, 'd.')
    assert sorted(enc.state_dict()) == sorted(e_ref) and sorted(gen.state_dict()) == sorted(d_ref)
    for k_, v in enc.state_dict().items():
        assert tuple(v.shape) == tuple(e_ref[k_].shape) and torch.equal(v, e_ref[k_]), k_
    for k_, v in gen.state_dict().items():
        assert tuple(v.shape) == tuple(d_ref[k_].shape) and torch.equal(v, d_ref[k_]), k_


def test_fourier_generator_names():
    import src.models as M
    fx = load_golden('step_mnist28_P16_fourier_normal')
    gen = M.SpatialGenerator(2, 64, num_layers=2, fourier_expansion=True, sigma=2.0 / 27)
    assert sorted(gen.state_dict()) == sorted(tdict(fx, 'd.'))
    fx = load_golden('decoder_resid')
    gen = M.SpatialGenerator(3, 64, num_layers=3, resid=True)
    assert sorted(gen.state_dict()) == sorted(tdict(fx, 'p.'))


def test_whole_module_pickle_roundtrip(tmp_path):
    """Checkpoints are whole-module pickles (reference train_mnist.py:672-681, src/utils.py:37-48) with classes at
    src.models.*; device tables must not leak into them."""
    import src.models as M
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(28, 1, 2, kernels_num=8, kernels_size=28,
                                                                    padding=8, groupconv=8, rot_refinement=True)
    enc.__dict__['_tb_cache'] = {'x': object()}
    p = tmp_path / 'inference.sav'
    torch.save(enc, p)
    raw = open(p, 'rb').read()
    assert b'src.models' in raw and b'InferenceNetwork_AttentionTranslation_AttentionRotation' in raw
    enc2 = torch.load(p, weights_only=False)
    assert '_tb_cache' not in enc2.__dict__
    for (k1, v1), (k2, v2) in zip(enc.state_dict().items(), enc2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    gen = M.SpatialGenerator(2, 16, num_layers=2, fourier_expansion=True, sigma=0.1)
    buf = io.BytesIO()
    torch.save(gen, buf)
    buf.seek(0)
    gen2 = torch.load(buf, weights_only=False)
    assert float(gen2.embed_latent.sigma) == pytest.approx(0.1)


def test_flat_adam_views_and_update_cpu():
    """FlatAdam bookkeeping (flat views, zero_grad, step counting) with a torch update function on CPU; the default
    update function is the HIP kernel (checked on the GPU in test_hip_primitives.py::test_adam_flat)."""
    from tvae import optim
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    calls = []

    def upd(p, g, m, v, step, lr, b1, b2, eps, scale):
        calls.append(step)
        O.adam_update([p], [g * scale], [m], [v], step, lr, b1, b2, eps)

    opt = optim.FlatAdam(ps, lr=1e-2, update_fn=upd)
    ropt = torch.optim.Adam(ref, lr=1e-2)
    assert ps[0].data_ptr() == opt.flat_p.data_ptr() and ps[1].grad.data_ptr() == opt.flat_g[64:].data_ptr()
    assert all(p.data_ptr() % 256 == opt.flat_p.data_ptr() % 256 for p in ps)
    for it in range(3):
        loss = sum(((p * (i + 1 + it)) ** 2).sum() for i, p in enumerate(ps))
        loss.backward()
        rl = sum(((p * (i + 1 + it)) ** 2).sum() for i, p in enumerate(ref))
        rl.backward()
        opt.step(); opt.zero_grad()
        ropt.step(); ropt.zero_grad()
        assert float(opt.flat_g.abs().sum()) == 0.0
    assert calls == [1, 2, 3]
    for p, r in zip(ps, ref):
        assert rel_err(p.detach(), r.detach()) < 1e-6
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode='max', factor=0.5, patience=0)
    sched.step(1.0); sched.step(0.0)
    assert opt.param_groups[0]['lr'] == pytest.approx(5e-3)


def test_flat_adam_takes_gradients_over_when_grads_are_none():
    """zero_grad(set_to_none=True), the training loops' form (round 4): the backward's AccumulateGrad nodes keep the gradient
    tensors (no add into a zeroed buffer); step() -- or the early bucket's hook, before it posts its all-reduce -- gathers them
    into the flat buffer.  Same trajectory as torch.optim.Adam; a parameter without a gradient gets a zero slice; the early
    bucket is posted with the gathered values."""
    from tvae import optim
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]

    def upd(p, g, m, v, step, lr, b1, b2, eps, scale):
        O.adam_update([p], [g * scale], [m], [v], step, lr, b1, b2, eps)

    class Reducer:
        active = True

        def __init__(self):
            self.begun = []

        def begin(self, seg):
            self.begun.append(seg.clone())

        def __call__(self, flat_g, start=0):
            return 1.0

    red = Reducer()
    opt = optim.FlatAdam(ps, lr=1e-2, update_fn=upd, reducer=red, early_params=2)
    ropt = torch.optim.Adam(ref, lr=1e-2)
    opt.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in ps)
    for it in range(3):
        use = ps if it != 1 else ps[:2]                  # step 1: the last parameter gets no gradient
        ruse = ref if it != 1 else ref[:2]
        sum(((p * (i + 1 + it)) ** 2).sum() for i, p in enumerate(use)).backward()
        sum(((p * (i + 1 + it)) ** 2).sum() for i, p in enumerate(ruse)).backward()
        assert len(red.begun) == it + 1                  # the early bucket (two parameters) was posted from inside the backward
        assert torch.equal(red.begun[-1][:12], ref[0].grad.reshape(-1)) and torch.equal(red.begun[-1][64:69], ref[1].grad)
        assert ps[0].grad.data_ptr() == opt.flat_g.data_ptr()
        opt.step()
        for p, r in zip(ps, ref):
            if r.grad is None:
                assert float(p.grad.abs().sum()) == 0.0
                r.grad = torch.zeros_like(r)             # (FlatAdam steps every parameter, with a zero gradient if it got none)
            else:
                assert torch.equal(p.grad, r.grad) and p.grad.data_ptr() >= opt.flat_g.data_ptr()
        assert float(opt.flat_g[12:64].abs().sum()) == 0.0           # padding never written
        ropt.step()
        opt.zero_grad(set_to_none=True)
        ropt.zero_grad(set_to_none=True)
    for p, r in zip(ps, ref):
        assert rel_err(p.detach(), r.detach()) < 1e-6


def test_resident_shard_plan_visits_every_image_once():
    """Shard-resident batches: every image exactly once per epoch, each rank only its own rows, global minibatch sizes
    known to every rank without communication."""
    from tvae import dp
    for n, gb, world in ((37, 16, 2), (100, 32, 8), (5, 6, 4), (64, 16, 1), (13, 6, 2), (60000, 100, 8),
                         (60000, 100, 3), (7, 2, 4), (1003, 100, 8), (9, 1, 4), (0, 4, 2)):
        plan = dp.shard_plan(n, gb, world)
        assert sum(sum(c) for c in plan) == n
        # the reference's loop: ceil(N / gb) minibatches, batch i holds min(gb, N - i*gb) images (train_mnist.py:586);
        # also when gb % world != 0 (driver default 100 over 8 ranks) and when gb < world
        assert len(plan) == (n + gb - 1) // gb
        for i, c in enumerate(plan):
            g = min(gb, n - i * gb)
            assert sum(c) == g
            assert max(c) - min(c) <= 1 or n - i * gb < gb + world      # even shares (the very tail may be ragged)
        for r in range(world):
            r0, r1 = dp.shard_bounds(n, r, world)
            assert sum(c[r] for c in plan) == r1 - r0
        glob = dp.resident_global_batches(n, gb, world, 3, 1)
        if n == 0:
            assert glob == []
            continue
        assert sorted(torch.cat(glob).tolist()) == list(range(n))
        assert [len(g) for g in glob] == [sum(c) for c in plan]
        assert all(len(g) <= gb for g in glob)
    assert dp.shard_bounds(10, 3, 4) == (8, 10) and dp.shard_bounds(10, 0, 4) == (0, 3)


def test_shard_slices_cover_every_minibatch():
    from tvae import dp
    for n, gb, world in ((10, 4, 2), (37, 8, 4), (5, 8, 2), (256, 256, 8)):
        per_rank = [list(dp.shard_slices(n, gb, r, world)) for r in range(world)]
        nb = (n + gb - 1) // gb
        assert all(len(p) == nb for p in per_rank)
        for b in range(nb):
            g = per_rank[0][b][2]
            idx = sorted(i for r in range(world) for i in range(per_rank[r][b][0], per_rank[r][b][1]))
            assert idx == list(range(b * gb, b * gb + g))
    assert torch.equal(dp.epoch_permutation(50, 3, 7), dp.epoch_permutation(50, 3, 7))
    assert not torch.equal(dp.epoch_permutation(50, 3, 7), dp.epoch_permutation(50, 3, 8))


def test_cli_flags_match_reference():
    """Every flag of the four reference scripts exists with the same option strings, default, choices and type
    (tests/golden/cli_flags.json is generated from the reference's own argparse objects)."""
    import json
    from tvae import driver
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'cli_flags.json')))
    for script, flags in ref.items():
        parser = driver.build_parser(script.replace('train_', ''))
        mine = {a.dest: a for a in parser._actions if a.dest != 'help'}
        for dest, spec in flags.items():
            assert dest in mine, (script, dest)
            a = mine[dest]
            assert list(a.option_strings) == spec['flags'], (script, dest)
            assert a.default == spec['default'], (script, dest, a.default, spec['default'])
            assert (list(a.choices) if a.choices else None) == spec['choices'], (script, dest)
            assert getattr(a.type, '__name__', None) == spec['type'], (script, dest)
            assert (a.nargs == 0) == spec['nargs0'], (script, dest)
        assert set(mine) - set(flags) == {'seed', 'synthetic', 'graph', 'no_graph'}, script      # this framework's own additions


def test_ctf_filter_matches_reference_golden():
    """src/ctf.py (product, host preprocessing) against filters generated by the reference's src/ctf.py."""
    import pandas as pd
    import src.ctf as C
    fx = load_golden('ctf_filters')
    cols = ['defocus', 'cs', 'voltage', 'apix', 'bfactor', 'ampcont', 'dfdiff', 'dfang']
    params = pd.DataFrame({c: fx[c] for c in cols})
    n = int(fx['n'])
    assert rel_err(C.ctf_filter(params, n, n), fx['filters']) < 1e-6


def test_mrc_codec_against_reference_written_files(tmp_path):
    """src/mrc.py reads stacks written by the reference codec (tests/golden/stack_ref*.mrc[s]) and writes
    byte-identical files; memory-mapped rank shards cover the stack exactly."""
    import src.mrc as mrc
    fx = load_golden('mrc_arrays')
    gdir = os.path.join(ROOT, 'tests', 'golden')
    raw = open(os.path.join(gdir, 'stack_ref.mrcs'), 'rb').read()
    arr, hdr, ext = mrc.parse(raw)
    assert (hdr.nx, hdr.ny, hdr.nz, hdr.mode, hdr.next) == (7, 6, 5, 2, 0) and ext == b''
    assert np.array_equal(arr, fx['stack']) and abs(hdr.xlen - 1.5) < 1e-6 and abs(hdr.rms - fx['stack'].std()) < 1e-6
    out = tmp_path / 'w.mrcs'
    with open(out, 'wb') as f:
        mrc.write(f, fx['stack'], ax=1.5, ay=2.5, az=3.5)
    assert open(out, 'rb').read() == raw                                   # byte-identical to the reference writer
    raw2 = open(os.path.join(gdir, 'stack_ref_ext.mrc'), 'rb').read()
    a2, h2, e2 = mrc.parse(raw2)
    assert h2.mode == 1 and h2.next == 16 and e2 == b'0123456789abcdef' and np.array_equal(a2, fx['single'])
    mm, h = mrc.open_stack(os.path.join(gdir, 'stack_ref.mrcs'))
    assert mm.shape == (5, 6, 7) and np.array_equal(np.asarray(mm), fx['stack'])
    parts = [mrc.read_shard(os.path.join(gdir, 'stack_ref.mrcs'), r, 2) for r in range(2)]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), fx['stack'])
    assert [p[1][:2] for p in parts] == [(0, 3), (3, 5)]


def test_driver_loads_only_its_shard_from_mrc_and_npy(tmp_path):
    """The particles driver opens MRC / .npy stacks lazily and reads, crops and normalises only the rows of its rank
    (dp.shard_bounds); the shards of two ranks tile what a single process loads, for both file formats."""
    import src.mrc as mrc
    from tvae import dp, driver
    g = np.random.default_rng(0)
    stack = g.normal(size=(11, 24, 24)).astype(np.float32)
    with open(tmp_path / 's.mrcs', 'wb') as f:
        mrc.write(f, stack)
    np.save(tmp_path / 's.npy', stack)
    for path in ('s.mrcs', 's.npy'):
        args = driver.build_parser('particles').parse_args(['--train-path', str(tmp_path / path), '--train-portion', '0.75',
                                                            '--crop', '20', '--normalize'])
        tr, te, name, n_raw, n_tr, n_te = driver._load_arrays('particles', args)
        assert (n_tr, n_te, n_raw) == (8, 3, 24) and tuple(tr.shape) == (8, 1, 20, 20) and tuple(te.shape) == (3, 1, 20, 20)
        ref = stack[:8, 2:22, 2:22]
        ref = (ref - ref.reshape(8, -1).mean(1)[:, None, None]) / ref.reshape(8, -1).std(1)[:, None, None]
        assert np.allclose(tr[:, 0].numpy(), ref, atol=1e-6)
        parts = [driver._load_arrays('particles', args, (r, 2)) for r in range(2)]
        assert torch.equal(torch.cat([p[0] for p in parts]), tr) and torch.equal(torch.cat([p[1] for p in parts]), te)
        assert [tuple(p[0].shape)[0] for p in parts] == [b - a for a, b in (dp.shard_bounds(8, r, 2) for r in range(2))]
        assert all(p[4:] == (8, 3) for p in parts)


def test_shape_validation_happens_before_any_launch():
    """Mis-shaped operands raise on the host (ValueError) without touching the GPU library."""
    from tvae import ops
    with pytest.raises(ValueError):
        ops.EncoderFn.apply(torch.zeros(2, 1, 8, 9), torch.zeros(4, 1, 1, 5, 5), torch.zeros(4), torch.zeros(4, 4),
                            torch.zeros(4), torch.zeros(7, 4), torch.zeros(7), 8, 1, 1)
    with pytest.raises(ValueError):
        ops.EncoderFn.apply(torch.zeros(2, 1, 8, 8), torch.zeros(4, 1, 1, 5, 5), torch.zeros(4), torch.zeros(4, 4),
                            torch.zeros(4), torch.zeros(7, 4), torch.zeros(7), 5, 1, 1)       # R not in {4,8,16}
    tb = type('T', (), dict(R=4, P=9))()
    with pytest.raises(ValueError):
        ops.HeadFn.apply(torch.zeros(7, 10), torch.zeros(2, 36), torch.zeros(2, 2), torch.zeros(2), tb, 2, 2)


def test_scratch_buffers_outlive_a_captured_graph():
    """once a hipGraph has been captured (tvae/graph.py -> ops.pin_scratch), growing a scratch buffer
    or the shared workspace must not free the block the graph still points to.  ADVICE r04: the pin belongs to the graph
    (a token), and releasing it (GraphedStep.close) lets the outgrown blocks go."""
    import torch
    from tvae import ops
    cpu = torch.device('cpu')
    saved = dict(ops._WS), dict(ops._PINNED)
    try:
        ops._WS.clear()
        ops._PINNED.clear()
        a = ops._scratch(cpu, 'k', 16)
        w = ops.workspace(cpu, 8)
        ops._scratch(cpu, 'k', 32)                       # before pinning: plain replacement
        assert not ops._PINNED
        b = ops._scratch(cpu, 'k', 32)
        tok = ops.pin_scratch()
        c = ops._scratch(cpu, 'k', 64)
        w2 = ops.workspace(cpu, 64)
        assert c.numel() >= 64 and w2.numel() == 64
        kept = {t.data_ptr() for t in ops._PINNED[tok]}
        assert b.data_ptr() in kept and w.data_ptr() in kept and a.data_ptr() not in kept
        assert ops._scratch(cpu, 'k', 8) is c            # no growth: same buffer, nothing new pinned
        assert len(ops._PINNED[tok]) == 2
        tok2 = ops.pin_scratch()                         # a second graph: both keep what is outgrown from now on
        ops._scratch(cpu, 'k', 128)
        assert len(ops._PINNED[tok]) == 3 and len(ops._PINNED[tok2]) == 1
        ops.unpin_scratch(tok)
        assert tok not in ops._PINNED and tok2 in ops._PINNED
        ops.unpin_scratch(tok2)
        assert not ops._PINNED
    finally:
        ops._WS.clear()
        ops._WS.update(saved[0])
        ops._PINNED.clear()
        ops._PINNED.update(saved[1])


def test_flat_adam_second_backward_raises():
    """with the two-bucket reducer active exactly one backward per step() is supported -- a second one reaching the
    early bucket after it was posted raises; after disable_early_bucket() (captured-graph path) the hooks are inert."""
    import pytest
    from tvae import optim

    class Reducer:
        active = True

        def begin(self, seg):
            pass

        def __call__(self, flat_g, start=0):
            return 1.0

    ps = [torch.nn.Parameter(torch.randn(3)), torch.nn.Parameter(torch.randn(2))]
    opt = optim.FlatAdam(ps, lr=1e-2, update_fn=lambda *a: None, reducer=Reducer(), early_params=1)
    opt.zero_grad(set_to_none=True)
    sum((p ** 2).sum() for p in ps).backward()
    with pytest.raises(RuntimeError, match='second backward'):
        sum((p ** 2).sum() for p in ps).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    opt.disable_early_bucket()
    for _ in range(2):                                   # hooks inert: gradient accumulation is allowed again
        sum((p ** 2).sum() for p in ps).backward()
    assert opt._early_seen == 0 and not opt._early_posted
    opt.step()


def test_hand_counted_kernels_use_no_scratch_and_no_packed_fp32():
    """the LDS-DMA ring kernels wait with hand-counted `s_waitcnt vmcnt(N)`; a compiler-emitted scratch spill (or
    any other vector-memory instruction it adds) would inflate the count and let a wave read LDS before its DMA has landed.
    Gate on the BUILT library: those kernels, and the lean epilogue instances whose point is a register budget, must have a
    private segment of 0 bytes; and no code object may contain packed-fp32 vector instructions (csrc/Makefile NOPK:
    profiles/README.md, rounds 3-4).  Reads the code objects embedded in libtvae_hip.so with llvm-readelf / llvm-objdump."""
    import re
    import subprocess
    import tempfile
    from tvae import _lib
    readelf, objdump = '/opt/rocm/lib/llvm/bin/llvm-readelf', '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not (os.path.exists(readelf) and os.path.exists(objdump) and os.path.exists(_lib.LIB_PATH)):
        import pytest
        pytest.skip('needs the ROCm llvm tools and the built library')
    data = open(_lib.LIB_PATH, 'rb').read()
    offs = [m.start() for m in re.finditer(b'\x7fELF\x02\x01\x01\x40', data)]
    assert len(offs) >= 20
    must_be_zero = re.compile(r'dft_out_ring_kernel|dft_dy_ring_kernel|dense_wgrad_x6_dma_kernel|dense_wgrad_x6_wide_kernel|enc_tail_wgrad_x6_kernel|'
                              r'dense_x6_kernelILi2ELi\dELi1E|dense_x6_kernelILi5ELi\dELi2E|dense_x6_plain4_kernelILi\dELi1E')
    seen, npk = 0, 0
    for o in offs:
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(data[o:])
            f.flush()
            notes = subprocess.run([readelf, '--notes', f.name], capture_output=True, text=True).stdout
            dis = subprocess.run([objdump, '-d', f.name], capture_output=True, text=True).stdout
        npk += len(re.findall(r'\bv_pk_(?:fma|mul|add)_f32\b', dis))
        for blk in re.split(r'\n\s+- ', notes):
            n = re.search(r'\.name:\s+(\S+)', blk)
            p = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk)
            if n and p and must_be_zero.search(n.group(1)):
                seen += 1
                assert int(p.group(1)) == 0, (n.group(1), int(p.group(1)))
    assert seen >= 23, seen
    assert npk == 0, npk
