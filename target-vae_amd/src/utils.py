"""Checkpointing helpers behind the reference's `src.utils` names (reference src/utils.py:7-48).

`EarlyStopping` keeps the constructor, the call signature, the two log messages and the `early_stop` /
`max_elbo` / `counter` attributes the drivers read.  Deliberate difference: the
reference moves the LIVE models to the CPU to pickle them and its driver moves them back every epoch; here a CPU
copy is pickled, because the live parameters are views into the flat optimiser buffer.  File names and the
whole-module pickle format (`generator.sav`, `inference.sav`, classes importable as src.models.*) are unchanged.
"""
import copy

import torch


def save_module_cpu(module, path):
    """Pickle a detached CPU copy of `module` in eval mode (whole-module pickle, like torch.save(model, path))."""
    clone = copy.deepcopy(module)
    for p in clone.parameters():
        p.grad = None
    torch.save(clone.cpu().eval(), path)


class EarlyStopping:
    """Tracks the best test ELBO: an epoch that beats it by at least `delta` checkpoints both models and resets the
    counter; `patience` epochs in a row without such an improvement set `early_stop`."""

    def __init__(self, patience=10, delta=0.01, save_path='./', digits=3):
        self.patience, self.delta = patience, delta
        self.save_path, self.digits = save_path, digits
        self.max_elbo = float('-inf')
        self.counter = 0
        self.early_stop = False

    def improved(self, elbo) -> bool:
        return not (elbo < self.max_elbo + self.delta)

    def __call__(self, elbo, encoder, generator, epoch):
        if self.improved(elbo):
            msg = self.save_checkpoint(elbo, encoder, generator, epoch)
            self.max_elbo, self.counter = elbo, 0
            return msg
        self.counter += 1
        if self.counter >= self.patience:
            self.early_stop = True
        return '#EarlyStopping counter: {} out of {}'.format(self.counter, self.patience)

    def save_checkpoint(self, elbo, encoder, generator, epoch):
        for module, stem in ((generator, 'generator.sav'), (encoder, 'inference.sav')):
            save_module_cpu(module, self.save_path + stem)
        return '#ELBO increased {:.4f}: --> {:.4f}.  Saving model ...'.format(self.max_elbo, elbo)
