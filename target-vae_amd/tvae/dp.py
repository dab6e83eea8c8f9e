"""Data parallelism for the TARGET-VAE step: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-device; every image is independent through encoder, sampling,
decoder and per-image ELBO and the loss is a batch mean (train_mnist.py:282,291), so the step shards
over images with ONE exchange: a sum all-reduce of the flat gradient buffer (1.5-12 MB).
On CPU-only boxes the same code runs over the `gloo` backend (tests/test_dp_gloo.py).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  A single process needs no process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # "nccl" IS RCCL on ROCm
        if backend == 'nccl':
            torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_buffers(*modules, src: int = 0, group=None):
    """Replicas must agree on every BUFFER too: RandomFourierEmbedding2d.weight / .bias are random buffers
    (src/models.py:37-39 of the reference), not parameters, so the flat-parameter broadcast does not cover them."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in modules:
        for b in m.buffers():
            dist.broadcast(b, src=src, group=group)


class GradReducer:
    """Sum all-reduce of the flat gradient buffer in (up to) two buckets; returns the scale Adam applies (1/world for
    equal shards).

    `begin(segment)` posts the all-reduce of a leading segment of the buffer asynchronously (the decoder's
    gradients are ready first, section 3.2): over RCCL the collective runs on the process group's own stream behind the
    kernels already queued on the compute stream, so it overlaps the encoder backward that follows; `__call__(flat_g,
    start)` reduces the rest and waits for the posted bucket.  Every rank issues the same two collectives in the same
    order (a rank with an empty shard issues both from the optimizer step), so the buckets always match up.

    `weight` handles a ragged last minibatch: rank r holds b_r images and its loss is a mean over b_r, so the
    global-batch gradient is sum_r (b_r / b_global) g_r; ranks pre-scale by b_r * world / b_global."""

    def __init__(self, group=None, always=False, abi=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # `always`: issue the collectives even in a one-rank group (they are identities there) -- lets a 1-GPU box
        # execute the RCCL path end to end (tests/test_dp_gpu.py)
        self.active = self.world > 1 or (bool(always) and dist.is_initialized())
        self.weight = 1.0
        self._pending = []
        self.posted_early = 0          # diagnostics: how many early buckets were posted from a backward
        # `abi` (EXPERIMENTAL, default off; TVAE_DP_ABI=1): the two buckets through the library's own entry point
        # tvae_allreduce_flat on a communicator created through the C ABI (include/tvae_hip.h) instead of torch.distributed's
        # all_reduce; the 128-byte id travels over the torch process group once.  Same collectives, same order, on the compute
        # stream.  It has only ever run with ONE rank (no multi-GPU node was available to any round).
        self._abi = None
        if abi is None:
            abi = os.environ.get('TVAE_DP_ABI', '0') == '1'
        if abi and self.active and dist.get_backend(group) == 'nccl':
            from ._lib import RcclComm
            dev = torch.device('cuda', torch.cuda.current_device())
            rank = dist.get_rank(group)
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(RcclComm.unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0, group=group)
            self._abi = RcclComm(self.world, bytes(idt.cpu().numpy().tobytes()), rank)

    def _abi_reduce(self, t: torch.Tensor) -> None:
        # On the COMPUTE stream itself: a second communicator on a stream of its own beside torch.distributed's
        # would leave the device-side order of the two communicators' collectives to the scheduler, rank by rank -- the classic
        # two-communicator deadlock.  Stream order = program order = the same on every rank; torch's own collectives wait
        # for the compute stream before they start.  (No overlap with the backward in this experimental mode.)
        self._abi.all_reduce_(t)

    def close(self) -> None:
        """Destroy the C-ABI communicator (if any) -- call before dist.destroy_process_group()."""
        if self._abi is not None:
            torch.cuda.synchronize()
            self._abi.close()
            self._abi = None

    def __del__(self):
        try:
            if getattr(self, '_abi', None) is not None:
                self._abi.close()
        except Exception:
            pass

    def set_local_fraction(self, local_b: int, global_b: int):
        self.weight = float(local_b) * self.world / float(global_b)

    def begin(self, segment: torch.Tensor) -> None:
        if not self.active:
            return
        if self.weight != 1.0:
            segment.mul_(self.weight)
        if self._abi is not None:
            self._abi_reduce(segment)
        else:
            self._pending.append(dist.all_reduce(segment, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.posted_early += 1

    def __call__(self, flat_g: torch.Tensor, start: int = 0) -> float:
        if not self.active:
            return 1.0
        rest = flat_g[start:] if start else flat_g
        if rest.numel() > 0:
            if self.weight != 1.0:
                rest.mul_(self.weight)
            if self._abi is not None:
                self._abi_reduce(rest)
            else:
                dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=self.group)
        for w in self._pending:
            w.wait()                                 # (torch Work objects and CUDA events both: the current stream waits)
        self._pending = []
        return 1.0 / self.world


def shard_slices(n_items: int, global_batch: int, rank: int, world: int):
    """Contiguous per-rank slice of every global minibatch: yields (lo, hi, b_global) index ranges into the
    epoch permutation.  Each global batch of size g is split as evenly as possible (first g % world ranks get
    one extra), so the union over ranks is exactly the reference's minibatch."""
    for start in range(0, n_items, global_batch):
        g = min(global_batch, n_items - start)
        base, extra = divmod(g, world)
        lo = start + rank * base + min(rank, extra)
        hi = lo + base + (1 if rank < extra else 0)
        yield lo, hi, g


def epoch_permutation(n_items: int, seed: int, epoch: int, shuffle: bool = True) -> torch.Tensor:
    """Same permutation on every rank (shared seed), like DataLoader(shuffle=True) with one generator."""
    if not shuffle:
        return torch.arange(n_items)
    gen = torch.Generator()
    gen.manual_seed(int(seed) * 1000003 + int(epoch))
    return torch.randperm(n_items, generator=gen)


class ShardedBatches:
    """Iterator of (y,) minibatch shards over a device-resident dataset tensor (the reference keeps the whole
    dataset on the device, train_mnist.py:495).  A ragged tail smaller than the number of ranks gives some ranks an
    EMPTY shard (b = 0): the iterator still yields it (reducer weight 0) and tvae.step.train_epoch / eval_model skip
    the forward/backward for it while still joining the gradient all-reduce and the Adam step."""

    def __init__(self, data, global_batch: int, rank: int = 0, world: int = 1, shuffle: bool = True,
                 seed: int = 0, reducer: GradReducer = None):
        # `data` is one tensor or a tuple of tensors indexed alike (e.g. images and their CTF filters)
        self.extra = tuple(data[1:]) if isinstance(data, (tuple, list)) else ()
        data = data[0] if isinstance(data, (tuple, list)) else data
        self.data, self.gb, self.rank, self.world = data, global_batch, rank, world
        self.shuffle, self.seed, self.reducer = shuffle, seed, reducer
        self.epoch = 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        return (self.data.shape[0] + self.gb - 1) // self.gb

    def __iter__(self):
        n = self.data.shape[0]
        perm = epoch_permutation(n, self.seed, self.epoch, self.shuffle).to(self.data.device)
        for lo, hi, g in shard_slices(n, self.gb, self.rank, self.world):
            if self.reducer is not None:
                self.reducer.set_local_fraction(hi - lo, g)
            idx = perm[lo:hi]
            yield (self.data.index_select(0, idx),) + tuple(t.index_select(0, idx) for t in self.extra)


# ---------------------------------------------------------------------------------------------
# per-rank RESIDENT shards: a rank keeps only its contiguous slice of the dataset on its GPU
# ---------------------------------------------------------------------------------------------
def shard_bounds(n_items: int, rank: int, world: int):
    """Rows [r0, r1) of the dataset that rank `rank` keeps resident (contiguous, sizes differ by at most one: the split
    src.mrc.read_shard makes on disk)."""
    base, extra = divmod(n_items, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def shard_plan(n_items: int, global_batch: int, world: int):
    """Batch plan of the shard-resident loop, identical on every rank without communication: counts[i][r] images of
    rank r's shard go into global minibatch i.

    The N images are laid out as one round-robin sequence of "slots" (round j holds one image of every rank whose
    shard has more than j rows: rank 0, 1, ..., world-1, then the next round) and global minibatch i takes slots
    [i*gb, min((i+1)*gb, N)).  So there are exactly ceil(N / gb) minibatches, minibatch i holds exactly
    g_i = min(gb, N - i*gb) images (the reference's loop, train_mnist.py:586-587), every rank's share of a minibatch
    is within one image of g_i / world, the ranks that get the extra image rotate from batch to batch when
    gb % world != 0, and every row of every shard is visited once per epoch (also when gb < world)."""
    if global_batch <= 0:
        raise ValueError('global_batch must be positive')
    base, extra = divmod(n_items, world)

    def taken(t, r):            # slots < t that belong to rank r
        if t <= base * world:
            return t // world + (1 if t % world > r else 0)
        return base + (1 if t - base * world > r else 0)

    plan = []
    for a in range(0, n_items, global_batch):
        b = min(a + global_batch, n_items)
        plan.append([taken(b, r) - taken(a, r) for r in range(world)])
    return plan


def local_permutation(n_local: int, seed: int, epoch: int, rank: int, shuffle: bool = True) -> torch.Tensor:
    if not shuffle:
        return torch.arange(n_local)
    gen = torch.Generator()
    gen.manual_seed((int(seed) * 1000003 + int(epoch)) * 8191 + int(rank) + 1)
    return torch.randperm(n_local, generator=gen)


def resident_global_batches(n_items: int, global_batch: int, world: int, seed: int, epoch: int, shuffle: bool = True):
    """The global minibatches of the shard-resident loop as dataset row indices (rank 0's rows first): what a single
    process must iterate over to reproduce a `world`-rank run (tests), and the definition of the sampling order --
    a shuffle stratified by shard: every global minibatch draws its even share from every rank's slice."""
    plan = shard_plan(n_items, global_batch, world)
    perms = [local_permutation(shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0], seed, epoch, r,
                               shuffle) + shard_bounds(n_items, r, world)[0] for r in range(world)]
    pos = [0] * world
    out = []
    for counts in plan:
        idx = []
        for r, c in enumerate(counts):
            idx.append(perms[r][pos[r]:pos[r] + c])
            pos[r] += c
        out.append(torch.cat(idx))
    return out


class ResidentShardBatches:
    """Iterator of (y,) minibatch shards over THIS RANK'S slice of the dataset only (`shard` = rows shard_bounds(..) of
    the full set, already on the device): nothing but the gradient all-reduce crosses ranks, and a GPU holds 1/world of
    the data.  With world = 1 this is the plain shuffled loop of the reference.  The plan (shard_plan) has the
    reference's ceil(N / gb) minibatches of gb images (last one ragged); a rank whose share of a minibatch is empty
    (gb < world, or the tail) still yields it, with reducer weight 0, as in ShardedBatches."""

    def __init__(self, shard, n_items: int, global_batch: int, rank: int = 0, world: int = 1, shuffle: bool = True,
                 seed: int = 0, reducer: GradReducer = None):
        self.extra = tuple(shard[1:]) if isinstance(shard, (tuple, list)) else ()
        self.data = shard[0] if isinstance(shard, (tuple, list)) else shard
        self.n_items, self.gb, self.rank, self.world = n_items, global_batch, rank, world
        self.shuffle, self.seed, self.reducer = shuffle, seed, reducer
        self.epoch = 0
        r0, r1 = shard_bounds(n_items, rank, world)
        if self.data.shape[0] != r1 - r0:
            raise ValueError(f'rank {rank} holds {self.data.shape[0]} rows, its shard of {n_items} has {r1 - r0}')
        self.plan = shard_plan(n_items, global_batch, world)

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        return len(self.plan)

    def local_count(self) -> int:
        return self.data.shape[0]

    def __iter__(self):
        perm = local_permutation(self.data.shape[0], self.seed, self.epoch, self.rank, self.shuffle).to(self.data.device)
        pos = 0
        for counts in self.plan:
            c = counts[self.rank]
            if self.reducer is not None:
                self.reducer.set_local_fraction(c, sum(counts))
            idx = perm[pos:pos + c]
            pos += c
            yield (self.data.index_select(0, idx),) + tuple(t.index_select(0, idx) for t in self.extra)


def allreduce_stats(values, device, group=None):
    """Sum a short list of python floats over ranks (logging scalars: sum b*elbo, sum b*err, sum b*kl, sum b)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(values)
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()
