"""Flat-buffer Adam for the MI355X hot path.

All parameters of generator + encoder (<= 2.9 M floats) live in ONE contiguous fp32 buffer,
their gradients in another; `p.data` / `p.grad` are views.  One fused HIP kernel (tvae_adam_flat)
replaces torch.optim.Adam's per-tensor loop (reference train_mnist.py:579,323-324) and one RCCL
all-reduce of the flat gradient buffer implements data parallelism (tvae/dp.py).
It subclasses torch.optim.Optimizer so that ReduceLROnPlateau (train_mnist.py:581) drives `lr` unchanged.
"""
from __future__ import annotations

import os

import torch

from . import ops


ALIGN = 64      # floats


class FlatAdam(torch.optim.Optimizer):
    """Adam over one flat parameter / gradient buffer (module docstring).

    Contract with a data-parallel reducer that posts an early bucket (`early_params`, two-bucket all-reduce): exactly ONE
    backward per `step()`.  The hook of the early parameters posts the all-reduce of their segment from inside that backward;
    a second backward before `step()` (gradient accumulation, a probe backward) would add into a buffer the collective is
    reducing, so it raises RuntimeError instead (tests/test_host_cpu.py::test_flat_adam_second_backward_raises).
    `TVAE_DP_EARLY=0` restores the single collective at `step()`, under which any number of backwards accumulate as with
    torch.optim.Adam.  `disable_early_bucket()` (tvae/graph.py: a captured backward must not post collectives) switches the
    hooks off for good."""

    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, reducer=None, update_fn=None, early_params=None):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._ps = [p for g in self.param_groups for p in g['params']]
        if len(self.param_groups) != 1:
            raise ValueError('FlatAdam keeps one parameter group (the reference uses one)')
        dev = self._ps[0].device
        # every parameter starts on a 256-B boundary of the flat buffer so that the float4 / aligned fast paths of
        # the GEMM loaders apply to parameter views too (padding stays zero: zero grad -> zero Adam update)
        self._offsets = []
        total = 0
        for p in self._ps:
            self._offsets.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self._gviews = []
        with torch.no_grad():
            for p, off in zip(self._ps, self._offsets):
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_p[off:off + n].view_as(p)
                gv = self.flat_g[off:off + n].view_as(p)
                p.grad = gv
                self._gviews.append(gv)
        self.steps = 0
        self.reducer = reducer            # callable(flat_g, start) -> grad_scale  (data-parallel all-reduce)
        self._update = update_fn or ops.adam_flat
        # Two gradient buckets: the first `early_params` parameters (the drivers pass the generator's --
        # the decoder backward runs first, section 3.2) form a leading segment of the flat buffer whose all-reduce is
        # posted from inside the backward, as soon as the last of them has accumulated its gradient; the rest follows at
        # step().  Only with a reducer that can post (`begin`) and more than one rank.
        self._early_n = 0
        self._early_end = 0
        self._early_seen = 0
        self._early_posted = False
        if (early_params and reducer is not None and hasattr(reducer, 'begin') and getattr(reducer, 'active', False)
                and os.environ.get('TVAE_DP_EARLY', '1') != '0'):
            self._early_n = min(int(early_params), len(self._ps))
            self._early_end = self._offsets[self._early_n] if self._early_n < len(self._ps) else total
            for p in self._ps[:self._early_n]:
                p.register_post_accumulate_grad_hook(self._early_hook)

    def disable_early_bucket(self) -> None:
        """No collective from inside the backward any more: step() reduces the whole buffer (the hooks stay registered but
        return at once, so nothing drifts: ADVICE r04)."""
        self._early_n = 0
        self._early_seen = 0
        self._early_posted = False

    def _early_hook(self, _p):
        if self._early_n == 0:
            return
        # Contract: ONE backward per optimizer step.  Once the early segment has been posted, the collective is reading and
        # writing flat_g[:early_end] on its own stream; a second backward before step() (gradient accumulation, a probe)
        # would accumulate into it underneath the all-reduce and produce wrong gradients without an error.
        if self._early_posted:
            raise RuntimeError('FlatAdam: a second backward reached the early gradient bucket before step(); with the '
                               'two-bucket all-reduce exactly one backward per step is supported (TVAE_DP_EARLY=0 '
                               'restores the single collective at step())')
        self._early_seen += 1
        if self._early_seen != self._early_n:
            return
        if any(p.grad is None for p in self._ps[:self._early_n]):
            return                          # a parameter of the bucket got no gradient: step() reduces everything
        self._gather(0, self._early_n)      # (gradients that were taken over instead of added in place: one multi-tensor copy)
        self._early_posted = True
        self.reducer.begin(self.flat_g[:self._early_end])

    def zero_grad(self, set_to_none: bool = False):
        """set_to_none=False: the flat gradient buffer is zeroed and every `p.grad` is its view of it (the next backward ADDS
        into the buffer: one elementwise launch per parameter).  set_to_none=True (the training loops): `p.grad = None`, so
        the next backward's AccumulateGrad nodes keep the gradient tensors their producers return, and step() (or the early
        bucket's hook) gathers them into the flat buffer with ONE multi-tensor copy -- 17 adds and the zero fill per step
        become one or two launches (round 4: profiles/tools/step_timeline.sh).  The buffer's padding is never written."""
        if set_to_none:
            for p in self._ps:
                p.grad = None
            return
        self.flat_g.zero_()
        for p, gv in zip(self._ps, self._gviews):
            p.grad = gv

    @torch.no_grad()
    def _gather(self, lo: int, hi: int) -> None:
        """Gradients of parameters lo .. hi-1 into their slices of the flat buffer; `p.grad` becomes the view again."""
        dst, src = [], []
        for p, gv in zip(self._ps[lo:hi], self._gviews[lo:hi]):
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                dst.append(gv)
                src.append(p.grad if p.grad.shape == gv.shape else p.grad.reshape(gv.shape))
            p.grad = gv
        if len(dst) == 1:
            dst[0].copy_(src[0])
        elif dst:
            torch._foreach_copy_(dst, src)

    @torch.no_grad()
    def step(self, closure=None):
        self._gather(self._early_n if self._early_posted else 0, len(self._ps))
        scale = 1.0
        if self.reducer is not None:
            if self._early_n and not self._early_posted:
                self.reducer.begin(self.flat_g[:self._early_end])      # no backward ran here (empty shard): same order
                self._early_posted = True
            scale = self.reducer(self.flat_g, self._early_end) if self._early_posted else self.reducer(self.flat_g)
            self._early_seen, self._early_posted = 0, False
        g = self.param_groups[0]
        self.steps += 1
        self._update(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.steps, g['lr'], g['betas'][0],
                     g['betas'][1], g['eps'], scale)
