"""Experiment driver shared by train_{mnist,particles,galaxy,dsprites}.py.

Keeps the reference command lines (flags and defaults of train_mnist.py:401-432, train_particles.py:481-523,
train_galaxy.py:401-431, train_dsprites.py:396-428), the stdout TSV log `Epoch Split ELBO Error KL`, the
`training_logs/<timestamp>_<dataset>_zDim_..._translation_..._rotation_...[_groupconvR]/` artefacts
(train_log.txt, generator.sav / inference.sav, *_epochNNN.sav; train_mnist.py:593-606,672-681) and the
optimiser policy (Adam 2e-4, ReduceLROnPlateau on test ELBO, EarlyStopping(20, 1e-4); :579-582,614).
New optional flags: --seed, --synthetic (synthetic data of the dataset's shape, no files needed), and
data parallelism through torchrun's RANK / WORLD_SIZE / LOCAL_RANK environment (one process per GPU, RCCL).
"""
from __future__ import annotations

import argparse
import datetime
import os
import sys

import numpy as np
import torch
import torch.nn as nn
from torch.optim.lr_scheduler import ReduceLROnPlateau

from . import dp, optim, step, tables

DEFAULTS = {
    #            kernel pad  in_ch image_dim gen_layers patience min_lr
    'mnist':    (28,    8,   1,    50,       2,         9,       0.0),
    'particles': (64,   16,  1,    None,     2,         9,       1e-6),
    'galaxy':   (64,    32,  3,    64,       4,         10,      0.0),
    'dsprites': (64,    32,  1,    64,       2,         9,       1e-6),     # train_dsprites.py:537
}


def build_parser(kind: str) -> argparse.ArgumentParser:
    ksz, pad, in_ch, img, glayers, _, _ = DEFAULTS[kind]
    titles = {'mnist': 'Train TARGET_VAE on MNIST/MNIST-N/MNIST-U datasets', 'particles': 'Training on particle datasets',
              'galaxy': 'Train TARGET-VAE on galaxy dataset', 'dsprites': 'Train TARGET-VAE on dSprites dataset'}
    p = argparse.ArgumentParser(titles[kind])
    if kind == 'mnist':
        p.add_argument('--dataset', choices=['mnist', 'mnist-U', 'mnist-N'], default='mnist-U')
    elif kind == 'galaxy':
        p.add_argument('--train-path', default='data/galaxy_zoo/galaxy_zoo_train.npy')
        p.add_argument('--test-path', default='data/galaxy_zoo/galaxy_zoo_test.npy')
    else:
        p.add_argument('--train-path', help='path to training data; or path to the whole data')
        p.add_argument('--test-path', help='path to testing data')
    if kind == 'particles':
        p.add_argument('--in-channels', type=int, default=1)
        p.add_argument('--ctf-train')
        p.add_argument('--ctf-test')
        p.add_argument('--scale', default=1, type=float)
    p.add_argument('-z', '--z-dim', type=int, default=2)
    p.add_argument('--t-inf', default='attention', choices=['unimodal', 'attention'])
    p.add_argument('--r-inf', default='attention+offsets', choices=['unimodal', 'attention', 'attention+offsets'])
    p.add_argument('--groupconv', type=int, default=8, choices=[0, 4, 8, 16])
    p.add_argument('--encoder-num-layers', type=int, default=2)
    p.add_argument('--encoder-kernel-number', type=int, default=128)
    p.add_argument('--encoder-kernel-size', type=int, default=ksz)
    p.add_argument('--encoder-padding', type=int, default=pad)
    if kind != 'particles':
        p.add_argument('--in-channels', type=int, default=in_ch)
        p.add_argument('--image-dim', type=int, default=img)
    p.add_argument('--fourier-expansion', action='store_true')
    p.add_argument('--generator-hidden-dim', type=int, default=512)
    p.add_argument('--generator-num-layers', type=int, default=glayers)
    p.add_argument('--generator-resid-layers', action='store_true')
    p.add_argument('--activation', choices=['tanh', 'leakyrelu'], default='leakyrelu')
    p.add_argument('-l', '--learning-rate', type=float, default=2e-4)
    p.add_argument('--minibatch-size', type=int, default=100)
    if kind == 'particles':
        p.add_argument('--train-portion', default=0.9, type=float)
    p.add_argument('--log-root', default='./training_logs')
    p.add_argument('--save-interval', default=20, type=int)
    p.add_argument('--num-epochs', type=int, default=500)
    p.add_argument('-d', '--device', type=int, default=0)
    if kind == 'particles':
        p.add_argument('--fit-noise', action='store_true')
        p.add_argument('--normalize', action='store_true')
        p.add_argument('--mask-radius', default=0, type=int)
        p.add_argument('--crop', default=0, type=int)
    # additions (do not change any reference flag)
    p.add_argument('--seed', type=int, default=None, help='seed for init / shuffling / noise (reference: unseeded)')
    p.add_argument('--synthetic', type=int, default=0, metavar='N',
                   help='train on N synthetic images of the dataset shape instead of loading files')
    # Round 6: ON BY DEFAULT wherever it applies (the attention branch without CTF filters / mask: every BASELINE configuration).
    # The minibatch shape of a run is fixed, the replay is bitwise the eager result
    # (tests/test_driver_gpu.py::test_graphed_step_bitwise_equals_eager) and the ~70 launches of a step are most of its cost at
    # small per-GPU batches (12 images: 1.68 -> 1.43 ms).  --graph insists (error where it cannot apply), --no-graph runs eagerly.
    p.add_argument('--graph', action='store_true',
                   help='insist on replaying a captured hipGraph of forward + backward for full-size minibatches '
                        '(tvae/graph.py); without this flag the graph is used wherever it applies')
    p.add_argument('--no-graph', action='store_true', help='run every step eagerly (no hipGraph capture)')
    return p


def _load_arrays(kind, args, shard=None):
    """Returns (train, test) float tensors shaped (N, Cin, n, n), the dataset name used in the log dir, the image side
    BEFORE --crop (the particles CTF kernels are sized from it, train_particles.py:540-577) and the sizes of the two
    FULL sets.  shard = (rank, world): the tensors hold only that rank's rows dp.shard_bounds(N, rank, world) of each
    set."""
    out = _load_arrays_full(kind, args, shard)
    if len(out) == 6:                                    # the loader sliced on disk already
        return out
    tr, te, name, n_raw = out
    n_tr, n_te = tr.shape[0], te.shape[0]
    if shard is not None:
        from . import dp
        (a0, a1), (b0, b1) = dp.shard_bounds(n_tr, *shard), dp.shard_bounds(n_te, *shard)
        tr, te = tr[a0:a1], te[b0:b1]
    return tr, te, name, n_raw, n_tr, n_te


def _load_arrays_full(kind, args, shard=None):
    if args.synthetic > 0:
        n = getattr(args, 'image_dim', None) or getattr(args, 'crop', 0) or 64    # particles: --crop sets the side
        cin = args.in_channels
        g = torch.Generator().manual_seed(0)
        mk = (lambda m: torch.randn(m, cin, n, n, generator=g)) if kind == 'particles' else \
            (lambda m: torch.rand(m, cin, n, n, generator=g))
        name = getattr(args, 'dataset', None) or ('synthetic' if kind == 'particles' else kind)
        return mk(args.synthetic), mk(max(args.synthetic // 10, 1)), name, n
    if kind == 'mnist':
        if args.dataset == 'mnist':
            try:
                import torchvision  # noqa: F401
            except ImportError as e:
                raise SystemExit('--dataset mnist needs torchvision (not installed here); use mnist-U / mnist-N '
                                 '(.npy) or --synthetic') from e
            import torchvision
            tr = torchvision.datasets.MNIST('data/', train=True, download=True)
            te = torchvision.datasets.MNIST('data/', train=False, download=True)
            to_np = lambda ds: np.stack([np.asarray(ds[i][0]) for i in range(len(ds))]).astype(np.uint8)
            a, b = to_np(tr), to_np(te)
        else:
            sub = {'mnist-U': 'mnist_U', 'mnist-N': 'mnist_N'}[args.dataset]
            a, b = np.load(f'data/{sub}/images_train.npy'), np.load(f'data/{sub}/images_test.npy')
        tr, te = torch.from_numpy(a).float() / 255, torch.from_numpy(b).float() / 255
        n = args.image_dim
        return tr.view(-1, args.in_channels, n, n), te.view(-1, args.in_channels, n, n), args.dataset, n
    if kind == 'galaxy':
        tr = torch.from_numpy(np.load(args.train_path)).float() / 255
        te = torch.from_numpy(np.load(args.test_path)).float() / 255
        n = args.image_dim                              # raw reinterpretation like the reference (train_galaxy.py:454)
        return tr.view(-1, args.in_channels, n, n), te.view(-1, args.in_channels, n, n), 'galaxy', n
    if kind == 'dsprites':
        tr = torch.from_numpy(np.load(args.train_path)[:1000]).float()      # reference truncation, train_dsprites.py:436
        te = torch.from_numpy(np.load(args.test_path)[:100]).float()
        n = args.image_dim
        return tr.view(-1, args.in_channels, n, n), te.view(-1, args.in_channels, n, n), 'dsprites', n
    # particles: .npy stacks or MRC/MRCS stacks (train_particles.py:454-461), opened LAZILY (np.load mmap / the MRC
    # memmap of src.mrc): with several ranks each one reads only its contiguous slice from disk (the split of
    # dp.shard_bounds = src.mrc.read_shard) before cropping / normalising, which are per-image operations
    def load(path):
        if path.endswith('.npy'):
            return np.load(path, mmap_mode='r')
        from src import mrc
        return mrc.open_stack(path)[0]
    if not args.train_path:
        raise SystemExit('please provide the train_path and/or test_path')
    if args.test_path:
        a, b = load(args.train_path), load(args.test_path)
    else:
        allim = load(args.train_path)
        k = int(allim.shape[0] * args.train_portion)
        a, b = allim[:k], allim[k:]
    n_raw = a.shape[-1]
    n_tr, n_te = a.shape[0], b.shape[0]
    if shard is not None:
        from . import dp
        (a0, a1), (b0, b1) = dp.shard_bounds(n_tr, *shard), dp.shard_bounds(n_te, *shard)
        a, b = a[a0:a1], b[b0:b1]
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    if args.crop > 0:
        def crop(s, m):
            n0 = s.shape[-1]
            lo = (n0 - m) // 2
            return s[:, lo:lo + m, lo:lo + m]
        a, b = crop(a, args.crop), crop(b, args.crop)
    if args.normalize:                                   # per-image mean/std (train_particles.py:592-600)
        def norm(s):
            f = s.reshape(s.shape[0], -1)
            return (s - f.mean(1)[:, None, None]) / f.std(1)[:, None, None]
        a, b = norm(a), norm(b)
    n = a.shape[-1]
    tr, te = torch.from_numpy(np.ascontiguousarray(a)).float(), torch.from_numpy(np.ascontiguousarray(b)).float()
    # log directory name: the training path with '/' -> '-' (train_particles.py:729-731)
    return (tr.view(-1, args.in_channels, n, n), te.view(-1, args.in_channels, n, n),
            args.train_path.replace('/', '-'), n_raw, n_tr, n_te)


def _load_ctf(args, n_train, n_test, n, shard=None):
    """Real-space CTF kernels per image (train_particles.py:540-577): odd size n-1 for even images, where n is the
    side of the stack as loaded, BEFORE --crop (the reference sizes the kernels first and crops afterwards).
    shard = (rank, world): only the filters of that rank's rows are built."""
    if not getattr(args, 'ctf_train', None):
        return None, None
    from src import ctf as C
    from . import dp
    kn = n - 1 if n % 2 == 0 else n
    if args.ctf_test:
        ptr, pte = C.parse_ctf(args.ctf_train), C.parse_ctf(args.ctf_test)
    else:
        allp = C.parse_ctf(args.ctf_train)
        ptr, pte = allp.iloc[:n_train].reset_index(drop=True), allp.iloc[n_train:].reset_index(drop=True)
    assert len(ptr) == n_train and len(pte) == n_test, 'one CTF parameter row per image is required'
    if shard is not None:
        (a0, a1), (b0, b1) = dp.shard_bounds(n_train, *shard), dp.shard_bounds(n_test, *shard)
        ptr, pte = ptr.iloc[a0:a1].reset_index(drop=True), pte.iloc[b0:b1].reset_index(drop=True)
    ftr, fte = C.ctf_filter(ptr, kn, kn, scale=args.scale), C.ctf_filter(pte, kn, kn, scale=args.scale)
    return torch.from_numpy(ftr).float().unsqueeze(1), torch.from_numpy(fte).float().unsqueeze(1)


def run(kind: str, argv=None):
    args = build_parser(kind).parse_args(argv)
    from src import models                     # drop-in classes (checkpoints pickle as src.models.*)
    from src.utils import EarlyStopping
    rank, world, local = dp.init_from_env()
    is_main = rank == 0
    num_epochs = args.num_epochs
    digits = int(np.log10(num_epochs)) + 1
    if args.seed is not None:
        torch.manual_seed(args.seed)
    y_train, y_test, dataset_name, n_raw, N, N_test = _load_arrays(kind, args, (rank, world))   # this rank's rows only
    image_dim = y_train.shape[-1]
    in_channels = y_train.shape[1]
    if not torch.cuda.is_available() or args.device == -1:
        raise SystemExit('the MI355X build has no CPU compute path (reference CPU mode -d -1 is not available)')
    dev_index = local if world > 1 else args.device
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if is_main:
        print('# using device:', device, f'(rank {rank}/{world})', file=sys.stderr)
    ctf_train = ctf_test = None
    if kind == 'particles' and not args.synthetic:
        ctf_train, ctf_test = _load_ctf(args, N, N_test, n_raw, (rank, world))
    # the arrays stay on the host here: each rank moves only its own slice to its GPU below (the reference keeps the whole
    # set on its one device, train_mnist.py:495)
    x_coord = torch.from_numpy(tables.image_coords(image_dim)).to(device)

    z_dim = args.z_dim
    activation = nn.Tanh if args.activation == 'tanh' else nn.LeakyReLU
    fourier_sigma = 2.0 / (image_dim - 1)                           # pixel size (train_mnist.py:511)
    n_out = {'mnist': 1, 'dsprites': 1, 'galaxy': 3}.get(kind, 2 if getattr(args, 'fit_noise', False) else 1)
    gen_kw = dict(n_out=n_out, num_layers=args.generator_num_layers, activation=activation,
                  resid=args.generator_resid_layers, fourier_expansion=args.fourier_expansion)
    if kind != 'dsprites':
        gen_kw['sigma'] = fourier_sigma                             # dsprites keeps the 0.01 default (quirk 8)
    generator_model = models.SpatialGenerator(z_dim, args.generator_hidden_dim, **gen_kw)

    t_inf, r_inf, group_conv = args.t_inf, args.r_inf, args.groupconv
    if kind == 'mnist' and args.dataset == 'mnist-N':
        theta_prior, normal_prior_over_r = np.pi / 4, True          # train_mnist.py:538-543
    elif kind == 'dsprites':
        # train_dsprites.py:509-523 sets normal_prior_over_r = False but never passes it to the encoder, so the
        # constructor default (True) applies: p(r) = Normal(0, pi) over the offsets
        theta_prior, normal_prior_over_r = np.pi, True
    else:
        theta_prior, normal_prior_over_r = np.pi, False
    if t_inf == 'unimodal' and r_inf == 'unimodal':                 # secondary encoders (train_mnist.py:546-557)
        encoder_model = models.InferenceNetwork_UnimodalTranslation_UnimodalRotation(
            image_dim * image_dim * in_channels, z_dim + 3, args.encoder_kernel_number,
            num_layers=args.encoder_num_layers, activation=activation)
    elif t_inf == 'attention' and r_inf == 'unimodal':
        encoder_model = models.InferenceNetwork_AttentionTranslation_UnimodalRotation(
            image_dim, in_channels, z_dim, kernels_num=args.encoder_kernel_number, activation=activation,
            groupconv=group_conv)
    elif t_inf == 'attention':
        if group_conv == 0:
            raise SystemExit('--groupconv 0 is not valid with attention over rotations (needs 4, 8 or 16)')
        encoder_model = models.InferenceNetwork_AttentionTranslation_AttentionRotation(
            image_dim, in_channels, z_dim, kernels_num=args.encoder_kernel_number,
            kernels_size=args.encoder_kernel_size, padding=args.encoder_padding, activation=activation,
            groupconv=group_conv, rot_refinement=(r_inf == 'attention+offsets'), theta_prior=theta_prior,
            normal_prior_over_r=normal_prior_over_r)
    else:
        raise SystemExit(f'--t-inf {t_inf} --r-inf {r_inf} is not a combination the reference supports')
    if kind == 'particles' and not (t_inf == 'attention' and r_inf != 'unimodal'):
        raise SystemExit('train_particles: the secondary inference branches are only wired for the BCE datasets')
    generator_model.to(device)
    encoder_model.to(device)
    if is_main:
        print(encoder_model)
        print(generator_model)

    params = list(generator_model.parameters()) + list(encoder_model.parameters())
    reducer = dp.GradReducer() if world > 1 else None
    optimizer = optim.FlatAdam(params, lr=args.learning_rate, reducer=reducer,
                               early_params=len(list(generator_model.parameters())))   # decoder gradients: first bucket
    if world > 1:
        torch.distributed.broadcast(optimizer.flat_p, src=0)       # identical replicas: parameters ...
        dp.broadcast_buffers(generator_model, encoder_model)       # ... and the random Fourier buffers
    # the shared seed covers init and the epoch permutation; the per-step noise (Exp(1), N(0,1)) comes from a
    # per-rank stream so that the shards of one global minibatch see independent draws
    if world > 1 or args.seed is not None:
        noise_gen = torch.Generator(device=device)
        noise_gen.manual_seed((args.seed if args.seed is not None else int(torch.initial_seed() % (1 << 31))) * world
                              + rank + 1)
        step.set_noise_generator(device, noise_gen)
    patience, min_lr = DEFAULTS[kind][5], DEFAULTS[kind][6]
    scheduler = ReduceLROnPlateau(optimizer, mode='max', factor=0.5, patience=patience, threshold=1e-4,
                                  threshold_mode='abs', cooldown=0, min_lr=min_lr, eps=1e-08)
    seed = args.seed if args.seed is not None else 0
    # every rank keeps only ITS contiguous slice of the two sets on its GPU
    to_dev = lambda t: None if t is None else t.to(device)
    train_src = to_dev(y_train) if ctf_train is None else (to_dev(y_train), to_dev(ctf_train))
    test_src = to_dev(y_test) if ctf_test is None else (to_dev(y_test), to_dev(ctf_test))
    del y_train, y_test, ctf_train, ctf_test
    train_it = dp.ResidentShardBatches(train_src, N, args.minibatch_size, rank, world, shuffle=True, seed=seed,
                                       reducer=reducer)
    test_it = dp.ResidentShardBatches(test_src, N_test, args.minibatch_size, rank, world, shuffle=False, seed=seed)
    likelihood = {'mnist': 'bce', 'dsprites': 'bce', 'galaxy': 'bce3'}.get(kind, 'gauss_var' if n_out == 2 else 'gauss')
    mask_radius = args.mask_radius if kind == 'particles' else None
    if kind == 'particles' and n_out == 2 and (isinstance(train_src, tuple) or args.mask_radius > 0):
        raise SystemExit('--fit-noise together with CTF filters or --mask-radius does not broadcast in the reference '
                         '(train_particles.py:303-307,330-333) and is not built')
    step_dim = args.encoder_padding if kind == 'particles' else image_dim   # reference positional argument
    graphed = None
    graph_ok = (t_inf == 'attention' and r_inf != 'unimodal') and not isinstance(train_src, tuple) and not (mask_radius or 0) > 0
    if args.graph and args.no_graph:
        raise SystemExit('--graph and --no-graph exclude each other')
    if args.graph and not graph_ok:
        raise SystemExit('--graph captures the TARGET-VAE attention branch without CTF filters / mask')
    if graph_ok and not args.no_graph and os.environ.get('TVAE_GRAPH', '1') != '0':
        from . import graph as _graph
        b_cap = train_it.plan[0][rank] if train_it.plan else 0
        if b_cap > 0:
            graphed = _graph.GraphedStep(x_coord, generator_model, encoder_model, optimizer, likelihood, b_cap,
                                         tuple(train_it.data.shape[1:]), device)

    output = sys.stdout
    log_file = None
    path_prefix = None
    if is_main:
        print('\t'.join(['Epoch', 'Split', 'ELBO', 'Error', 'KL']), file=output)
        os.makedirs(args.log_root, exist_ok=True)
        desc = '_'.join([datetime.datetime.now().strftime('%Y-%m-%d-%H-%M'), dataset_name, 'zDim', str(z_dim),
                         'translation', t_inf, 'rotation', r_inf])
        if group_conv > 0:
            desc += '_groupconv' + str(group_conv)
        if kind == 'particles':                                     # train_particles.py:734-737
            if getattr(args, 'ctf_train', None):
                desc += '_ctf'
            if args.fourier_expansion:
                desc += '_Fr_sigma' + str(fourier_sigma)
        path_prefix = os.path.join(args.log_root, desc, '')
        os.makedirs(path_prefix, exist_ok=True)
        print('# learning-rate is {}'.format(args.learning_rate))
        log_file = open(path_prefix + 'train_log.txt', 'w', 1)
        print(desc + '\n', file=log_file)
        print('\n\nargs:', file=log_file)
        print(str(args), file=log_file)
        print('\nEncoder model: \n {}'.format(encoder_model), file=log_file)
        print('\nGenerator model: \n {}'.format(generator_model), file=log_file)
        print('\n\n', file=log_file)
        print('\t'.join(['Epoch', 'Split', 'ELBO', 'Error', 'KL']) + '\n', file=log_file)
    early_stopping = EarlyStopping(patience=20, delta=1e-4, save_path=path_prefix or './', digits=digits)

    def emit(line, blank=False):
        if is_main:
            print(line, file=output)
            print(line, file=log_file)
            if blank:
                print('\n', file=output)
                print('\n', file=log_file)

    def global_means(e, err, kl, count):
        """Batch-size weighted means over all ranks (each rank's running means are weighted by its image count)."""
        s = dp.allreduce_stats([e * count, err * count, kl * count, float(count)], device)
        return s[0] / s[3], s[1] / s[3], s[2] / s[3]

    for epoch in range(num_epochs):
        train_it.set_epoch(epoch)
        n_local = train_it.local_count()
        e, err, kl = step.train_epoch(train_it, x_coord, generator_model, encoder_model, optimizer, t_inf, r_inf, epoch,
                                      num_epochs, max(n_local, 1), device, params, theta_prior, group_conv, step_dim,
                                      likelihood=likelihood, progress=is_main, mask_radius=mask_radius, graphed=graphed)
        e, err, kl = global_means(e, err, kl, n_local)
        emit('\t'.join([str(epoch + 1), 'train', str(e), str(err), str(kl)]))
        n_test = test_it.local_count()
        e, err, kl = step.eval_model(test_it, x_coord, generator_model, encoder_model, t_inf, r_inf, epoch, device,
                                     theta_prior, group_conv, step_dim, likelihood=likelihood, mask_radius=mask_radius)
        e, err, kl = global_means(e, err, kl, n_test)
        emit('\t'.join([str(epoch + 1), 'test', str(e), str(err), str(kl)]))
        if is_main:
            msg = early_stopping(e, encoder_model, generator_model, epoch + 1)
            emit(msg, blank=True)
        stop = early_stopping.early_stop
        if world > 1:                                  # rank 0 decides; one int broadcast
            flag = torch.tensor([int(stop)], device=device)
            torch.distributed.broadcast(flag, src=0)
            stop = bool(flag.item())
        if stop:
            if is_main:
                print('*** Early stopping ***')
            break
        scheduler.step(e)                              # same test ELBO on every rank -> same lr everywhere
        if is_main and (epoch + 1) % args.save_interval == 0:
            from src.utils import save_module_cpu
            tag = str(epoch + 1).zfill(digits)
            save_module_cpu(generator_model, path_prefix + 'generator_epoch{}.sav'.format(tag))
            save_module_cpu(encoder_model, path_prefix + 'inference_epoch{}.sav'.format(tag))
    if log_file is not None:
        log_file.close()
    if world > 1:
        if reducer is not None:
            reducer.close()                              # (a C-ABI communicator dies before the process group: ADVICE r05)
        torch.distributed.destroy_process_group()
