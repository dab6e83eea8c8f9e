"""hipGraph replay of the forward + backward of the training step (opt-in: `--graph` of the drivers, `bench.py --graph`).

The 28x28 workloads launch ~70 kernels per 4 ms step, so the host's launch cost shows; capturing forward + backward once and
replaying it removes it.  What stays outside the graph: the minibatch copy into a static buffer, the three noise draws
(torch RNG kernels, the reference's order: train_mnist.py:206,230 / src/models.py:387), the gradient all-reduce and the
fused Adam launch -- so learning-rate changes, data parallelism and the optimizer state behave exactly as in eager mode.

Replay is BIT FOR BIT equal to eager execution (tests/test_driver_gpu.py::test_graphed_step_bitwise_equals_eager, also
with a host synchronize or a `deepcopy(model).cpu()` between replays: the two triggers under which round 2 saw corrupted
replays no longer reproduce with this library -- profiles/tools/graph_replay_probe.py, profiles/README.md round 3).
A minibatch of another size than the captured one (the ragged tail of an epoch) runs eagerly.
"""
from __future__ import annotations

import torch

from . import step as _step


class GraphedStep:
    def __init__(self, x_coord, generator_model, encoder_model, optim, likelihood, batch, image_shape, device):
        self.x, self.gen, self.enc, self.opt, self.lik = x_coord, generator_model, encoder_model, optim, likelihood
        self.B = int(batch)
        dev = torch.device(device)
        enc = encoder_model
        self.rp, self.zd = enc.groupconv * enc.output_size() ** 2, enc.latent_dim
        self.y = torch.zeros((self.B,) + tuple(image_shape), dtype=torch.float32, device=dev)
        self.E = torch.ones(self.B, self.rp, dtype=torch.float32, device=dev)
        self.ez = torch.zeros(self.B, self.zd, dtype=torch.float32, device=dev)
        self.et = torch.zeros(self.B, dtype=torch.float32, device=dev)
        optim.disable_early_bucket()       # no collective from inside a captured backward: step() reduces everything
        _step.pixel_spacing(x_coord)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):      # warm-up on the capture stream: every scratch buffer and table exists afterwards
            for _ in range(2):
                self._fwd_bwd()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.terms = self._fwd_bwd()
        torch.cuda.synchronize(dev)
        # the graph now holds raw pointers into the scratch buffers of tvae.ops: from here on an outgrown buffer is kept
        # alive instead of freed (a larger eager batch later must not hand the graph's blocks back to the allocator)
        from . import ops as _ops
        self._pin = _ops.pin_scratch()
        optim.flat_g.zero_()

    def close(self) -> None:
        """Drop the graph and release the scratch blocks that were only kept alive for its raw pointers (the
        pinned list used to be process-global and grew for ever)."""
        self.graph = None
        if getattr(self, '_pin', None) is not None:
            from . import ops as _ops
            _ops.unpin_scratch(self._pin)
            self._pin = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _fwd_bwd(self):
        self.opt.flat_g.zero_()
        for p, gv in zip(self.opt._ps, self.opt._gviews):
            p.grad = gv
        elbo, log_p, kl = _step.elbo_terms(self.x, self.y, self.gen, self.enc, self.lik, (self.E, self.ez, self.et))
        _step.backward_neg_elbo(elbo)
        return torch.stack([elbo.detach().double(), log_p.detach().double(), kl.detach().double()])

    def draw_noise(self, generator=None):
        """The reference's three draws, in its order, into the static buffers."""
        if generator is None and _step._NOISE_GEN:
            generator = _step._NOISE_GEN.get((self.y.device.type, self.y.device.index))
        self.E.exponential_(generator=generator)
        self.ez.normal_(generator=generator)
        self.et.normal_(generator=generator)

    def run(self, y, noise=None):
        """Forward + backward of one minibatch of the captured size; gradients land in the optimizer's flat buffer.
        Returns the (elbo, log_p, kl) tensor (float64, 3 values) of this replay."""
        self.y.copy_(y.reshape(self.y.shape))
        if noise is None:
            self.draw_noise()
        else:
            self.E.copy_(noise[0].reshape(self.E.shape))
            self.ez.copy_(noise[1].reshape(self.ez.shape))
            self.et.copy_(noise[2].reshape(self.et.shape))
        self.graph.replay()
        for p, gv in zip(self.opt._ps, self.opt._gviews):      # the replay wrote the flat buffer: step() must not gather
            p.grad = gv
        return self.terms
