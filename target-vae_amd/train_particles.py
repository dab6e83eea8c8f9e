#!/usr/bin/env python3
"""Drop-in for the reference train_particles.py (same flags, log lines and checkpoint artefacts) on the MI355X hot path.
Step functions with the reference signatures live in tvae.step; see tvae/driver.py for the epoch driver."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from tvae.driver import run
from tvae.step import eval_model, train_epoch  # noqa: F401  (reference module-level names)
from tvae.step import eval_minibatch_particles as eval_minibatch  # noqa: F401,E402


def main():
    run('particles')


if __name__ == '__main__':
    main()
