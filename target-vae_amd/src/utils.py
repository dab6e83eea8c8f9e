"""EarlyStopping with the reference interface (reference src/utils.py:7-48).

Difference kept deliberately: the reference moves the LIVE models to the CPU to
pickle them and the driver moves them back every epoch; here a CPU copy is pickled instead, because the live
parameters are views into the flat optimiser buffer.  File names and the whole-module pickle format
(`generator.sav`, `inference.sav`, classes importable as src.models.*) are unchanged.
"""
import copy

import numpy as np
import torch


def save_module_cpu(module, path):
    clone = copy.deepcopy(module)
    for p in clone.parameters():
        p.grad = None
    torch.save(clone.cpu().eval(), path)


class EarlyStopping:
    """Stops training when the test ELBO has not improved by `delta` for `patience` epochs; checkpoints the best."""

    def __init__(self, patience=10, delta=0.01, save_path='./', digits=3):
        self.patience = patience
        self.counter = 0
        self.max_elbo = -np.inf
        self.early_stop = False
        self.delta = delta
        self.save_path = save_path
        self.digits = digits

    def __call__(self, elbo, encoder, generator, epoch):
        if elbo < self.max_elbo + self.delta:
            self.counter += 1
            msg = '#EarlyStopping counter: {} out of {}'.format(self.counter, self.patience)
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            msg = self.save_checkpoint(elbo, encoder, generator, epoch)
            self.max_elbo = elbo
            self.counter = 0
        return msg

    def save_checkpoint(self, elbo, encoder, generator, epoch):
        msg = '#ELBO increased {:.4f}: --> {:.4f}.  Saving model ...'.format(self.max_elbo, elbo)
        save_module_cpu(generator, self.save_path + 'generator.sav')
        save_module_cpu(encoder, self.save_path + 'inference.sav')
        return msg
