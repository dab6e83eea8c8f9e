"""The four command-line drivers and eval_model on the GPU.

Reference behaviour pinned here: stdout TSV `Epoch Split ELBO Error KL` with one train and one test line per epoch
(train_mnist.py:590,640-664), the `training_logs/<timestamp>_<dataset>_zDim_<z>_translation_<t>_rotation_<r>[_groupconvR]`
directory with train_log.txt, generator.sav / inference.sav (best test ELBO, src/utils.py:37-48) and
generator_epochNNN.sav / inference_epochNNN.sav every --save-interval epochs (train_mnist.py:672-681), all of them
whole-module pickles that reload as src.models.* classes."""
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, load_golden, tdict
from oracle import tvae_oracle as O

pytestmark = pytest.mark.gpu

SMALL = ['--minibatch-size', '8', '--num-epochs', '2', '--save-interval', '1', '--encoder-kernel-number', '8',
         '--generator-hidden-dim', '32', '--encoder-kernel-size', '20', '--encoder-padding', '4', '--seed', '3',
         '--synthetic', '20']
CASES = {
    'train_mnist': (['--dataset', 'mnist-U', '--image-dim', '20', '-z', '2'], r'mnist-U_zDim_2'),
    'train_particles': (['--crop', '20', '-z', '2', '--mask-radius', '7', '--fourier-expansion'],
                        r'synthetic_zDim_2'),
    'train_galaxy': (['--image-dim', '20', '-z', '5', '--groupconv', '4'], r'galaxy_zDim_5'),
    'train_dsprites': (['--image-dim', '20', '-z', '2', '--r-inf', 'attention'], r'dsprites_zDim_2'),
}


@pytest.mark.parametrize('script', sorted(CASES))
def test_cli_two_epochs(script, tmp_path):
    extra, stem = CASES[script]
    cmd = [sys.executable, os.path.join(PKG, script + '.py')] + SMALL + extra + ['--log-root', str(tmp_path / 'logs')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [ln.split('\t') for ln in r.stdout.splitlines() if '\t' in ln]
    assert rows[0] == ['Epoch', 'Split', 'ELBO', 'Error', 'KL']
    body = rows[1:]
    assert [(b[0], b[1]) for b in body] == [('1', 'train'), ('1', 'test'), ('2', 'train'), ('2', 'test')]
    vals = np.array([[float(v) for v in b[2:]] for b in body])
    assert np.isfinite(vals).all()
    assert np.allclose(vals[:, 0], -vals[:, 1] - vals[:, 2], rtol=1e-6)           # ELBO = -(Error) - KL
    assert '#ELBO increased' in r.stdout                                            # src/utils.py:40 message
    dirs = glob.glob(str(tmp_path / 'logs' / '*'))
    assert len(dirs) == 1
    name = os.path.basename(dirs[0])
    gc = '4' if script == 'train_galaxy' else '8'
    r_inf = 'attention' if script == 'train_dsprites' else 'attention\\+offsets'
    pat = r'^\d{4}-\d{2}-\d{2}-\d{2}-\d{2}_' + stem + '_translation_attention_rotation_' + r_inf + '_groupconv' + gc
    if script == 'train_particles':
        pat += r'_Fr_sigma' + re.escape(str(2.0 / 19))                              # train_particles.py:736-737
    assert re.match(pat + '$', name), name
    files = sorted(os.listdir(dirs[0]))
    assert files == ['generator.sav', 'generator_epoch1.sav', 'generator_epoch2.sav', 'inference.sav',
                     'inference_epoch1.sav', 'inference_epoch2.sav', 'train_log.txt'], files
    log = open(os.path.join(dirs[0], 'train_log.txt')).read()
    assert log.startswith(name) and 'Epoch\tSplit\tELBO\tError\tKL' in log and '2\ttest\t' in log
    # whole-module pickles reload as the drop-in classes and evaluate on the device
    sys.path.insert(0, PKG)
    import src.models as M
    from tvae import step, tables
    gen = torch.load(os.path.join(dirs[0], 'generator.sav'), weights_only=False)
    enc = torch.load(os.path.join(dirs[0], 'inference.sav'), weights_only=False)
    assert type(gen) is M.SpatialGenerator and type(enc) is M.InferenceNetwork_AttentionTranslation_AttentionRotation
    assert not gen.training and not enc.training and next(gen.parameters()).device.type == 'cpu'
    dev = torch.device('cuda:0')
    gen, enc = gen.to(dev), enc.to(dev)
    cin = 3 if script == 'train_galaxy' else 1
    y = torch.rand(6, cin, 20, 20, device=dev)
    x = torch.from_numpy(tables.image_coords(20)).to(dev)
    lik = {'train_galaxy': 'bce3', 'train_particles': 'gauss'}.get(script, 'bce')
    kw = dict(mask_radius=7) if script == 'train_particles' else {}
    e, err, kl = step.eval_model([(y[:4],), (y[4:],)], x, gen, enc, 'attention',
                                 'attention' if script == 'train_dsprites' else 'attention+offsets', 0, dev, np.pi,
                                 int(gc), 4 if script == 'train_particles' else 20, likelihood=lik, **kw)
    assert np.isfinite([e, err, kl]).all() and abs(e + err + kl) < 1e-6 * abs(e)


def test_particles_cli_from_mrc_stack(tmp_path):
    """MRC/MRCS stack -> device-resident dataset: train_particles on a stack written by src.mrc.write
    (byte-identical to the reference writer, tests/test_host_cpu.py), split by --train-portion, cropped and normalised,
    two epochs; the same run from the .npy copy of the stack gives the same TSV numbers."""
    import src.mrc as mrc
    g = np.random.default_rng(1)
    stack = (g.normal(size=(24, 22, 22)) * 3 + 1).astype(np.float32)
    with open(tmp_path / 'p.mrcs', 'wb') as f:
        mrc.write(f, stack)
    np.save(tmp_path / 'p.npy', stack)
    outs = []
    for path in ('p.mrcs', 'p.npy'):
        cmd = [sys.executable, os.path.join(PKG, 'train_particles.py'), '--train-path', str(tmp_path / path),
               '--train-portion', '0.75', '--crop', '20', '--normalize', '-z', '2', '--minibatch-size', '6', '--num-epochs',
               '2', '--encoder-kernel-number', '8', '--generator-hidden-dim', '32', '--encoder-kernel-size', '20',
               '--encoder-padding', '4', '--seed', '3', '--log-root', str(tmp_path / ('logs_' + path))]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-3000:]
        rows = [ln.split('\t') for ln in r.stdout.splitlines() if '\t' in ln]
        assert rows[0] == ['Epoch', 'Split', 'ELBO', 'Error', 'KL'] and len(rows) == 5
        outs.append(np.array([[float(v) for v in b[2:]] for b in rows[1:]]))
        assert np.isfinite(outs[-1]).all()
    assert np.allclose(outs[0], outs[1], rtol=1e-5), (outs[0], outs[1])


def test_eval_model_matches_oracle():
    """eval_model (train_mnist.py:352-387): no-grad twin of train_epoch -- batch-weighted running means of the three
    ELBO terms over a ragged pair of minibatches, against the oracle with the same injected noise; parameters and
    gradients untouched, modules left in eval mode."""
    from tvae import step
    import src.models as M
    fx = load_golden('epoch_2steps')
    n, cin, zd, C, k, p, R, refine, normal, hid, L, n_out, fourier, resid = [int(v) for v in fx['cfg']]
    dev = torch.device('cuda:0')
    enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
        n, cin, zd, kernels_num=C, kernels_size=k, padding=p, groupconv=R, rot_refinement=True,
        theta_prior=float(fx['theta_prior']), normal_prior_over_r=False)
    gen = M.SpatialGenerator(zd, hid, n_out=n_out, num_layers=L)
    enc.load_state_dict(tdict(fx, 'e.'))
    gen.load_state_dict(tdict(fx, 'd.'))
    data = torch.from_numpy(fx['data'])
    sizes = [(0, 4), (4, 7)]                     # ragged: 4 + 3 images
    noises = [tuple(torch.from_numpy(fx[f'{k_}{i}'])[:hi - lo] for k_ in ('E', 'eps_z', 'eps_theta'))
              for i, (lo, hi) in enumerate(sizes)]
    want = np.zeros(3)
    c = 0
    for (lo, hi), (E, ez, et) in zip(sizes, noises):
        with torch.no_grad():
            e, lp, kl = O.elbo_step(O.image_coords(n), data[lo:hi], enc.state_dict(), gen.state_dict(), R=R, padding=p,
                                    rot_refinement=True, theta_prior=float(fx['theta_prior']), normal_prior_over_r=False,
                                    num_layers=L, likelihood='bce', E=E, eps_z=ez, eps_theta=et)
        c += hi - lo
        want += (hi - lo) * (np.array([float(e), -float(lp), float(kl)]) - want) / c
    enc, gen = enc.to(dev), gen.to(dev)
    before = [t.detach().clone() for t in list(enc.parameters()) + list(gen.parameters())]
    it = [(data[lo:hi].to(dev),) for lo, hi in sizes]
    got = step.eval_model(it, O.image_coords(n).to(dev), gen, enc, 'attention', 'attention+offsets', 0, dev, np.pi, R,
                          n, noise_iter=iter([tuple(t.to(dev) for t in nz) for nz in noises]))
    assert np.allclose(got, want, rtol=1e-4), (got, want)
    assert not enc.training and not gen.training
    for t, b in zip(list(enc.parameters()) + list(gen.parameters()), before):
        assert t.grad is None and torch.equal(t.detach(), b)


def test_graphed_step_bitwise_equals_eager():
    """tvae.graph.GraphedStep (opt-in `--graph`): replaying the captured forward + backward gives BIT FOR BIT the eager
    gradients and ELBO terms over several optimizer steps -- also with the two triggers under which round 2 saw corrupted
    replays (a host synchronize between a replay and the next launch; deepcopy(model).cpu() between replays)."""
    import copy
    sys.path.insert(0, PKG)
    import src.models as M
    from tvae import graph, optim, step, tables
    dev = torch.device('cuda:0')
    n, R, B = 28, 8, 32

    def models():
        torch.manual_seed(5)
        gen = M.SpatialGenerator(2, 512, num_layers=2).to(dev)
        enc = M.InferenceNetwork_AttentionTranslation_AttentionRotation(
            n, 1, 2, kernels_num=128, kernels_size=28, padding=8, groupconv=R, rot_refinement=True, theta_prior=np.pi,
            normal_prior_over_r=False).to(dev)
        params = list(gen.parameters()) + list(enc.parameters())
        return gen, enc, params, optim.FlatAdam(params, lr=1e-3)
    x = torch.from_numpy(tables.image_coords(n)).to(dev)
    g = torch.Generator(device=dev).manual_seed(3)
    ys = [torch.rand(B, 1, n, n, device=dev, generator=g) for _ in range(4)]
    noises = [step.draw_noise(B, R * 17 * 17, 2, dev, generator=g) for _ in range(4)]
    # eager trajectory
    gen, enc, params, opt = models()
    eager = []
    for y, nz in zip(ys, noises):
        e, lp, kl = step.elbo_terms(x, y, gen, enc, 'bce', nz)
        (-e).backward()
        eager.append((opt.flat_g.clone(), torch.stack([e.detach().double(), lp.detach().double(), kl.detach().double()])))
        opt.step()
        opt.zero_grad()
    p_eager = opt.flat_p.clone()
    # the same trajectory through the graph
    gen, enc, params, opt = models()
    gs = graph.GraphedStep(x, gen, enc, opt, 'bce', B, (1, n, n), dev)
    for i, (y, nz) in enumerate(zip(ys, noises)):
        terms = gs.run(y, nz)
        if i == 1:
            torch.cuda.synchronize()
        if i == 2:
            copy.deepcopy(gen).cpu()
        assert torch.equal(opt.flat_g.view(torch.int32), eager[i][0].view(torch.int32)), i
        assert torch.equal(terms, eager[i][1]), i
        opt.step()
        opt.zero_grad()
    assert torch.equal(opt.flat_p, p_eager)
    # train_epoch takes the replay for minibatches of the captured size and the eager path for a ragged tail
    data = torch.cat(ys + [ys[0][:5]])
    it = [(data[i:i + B],) for i in range(0, data.shape[0], B)]
    gen, enc, params, opt = models()
    gs = graph.GraphedStep(x, gen, enc, opt, 'bce', B, (1, n, n), dev)
    nz_all = noises + [tuple(t[:5] for t in noises[0])]
    r1 = step.train_epoch(it, x, gen, enc, opt, 'attention', 'attention+offsets', 0, 1, data.shape[0], dev, params, np.pi, R, n,
                          progress=False, noise_iter=iter(nz_all), graphed=gs)
    p1 = opt.flat_p.clone()
    gen, enc, params, opt = models()
    r0 = step.train_epoch(it, x, gen, enc, opt, 'attention', 'attention+offsets', 0, 1, data.shape[0], dev, params, np.pi, R, n,
                          progress=False, noise_iter=iter(nz_all))
    assert r0 == r1 and torch.equal(opt.flat_p, p1)


def test_cli_graph_flag_same_log_as_eager(tmp_path):
    """`train_mnist.py` with the hipGraph step (the default since round 6, and with --graph insisting on it) prints the same
    TSV lines as the eager run (--no-graph) under the same seed."""
    outs = []
    for extra in (['--no-graph'], [], ['--graph']):
        cmd = [sys.executable, os.path.join(PKG, 'train_mnist.py')] + SMALL + CASES['train_mnist'][0] + extra + \
            ['--seed', '7', '--log-root', str(tmp_path / ('logs' + ''.join(extra)))]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append([ln for ln in r.stdout.splitlines() if '\t' in ln])
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) == 5
