import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_batch(rng, B, W, T, K1, V, N_u, pad_hist_frac=0.3):
    """Synthetic id-form batch with ragged (zero-padded) titles and front-padded histories."""
    import numpy as np

    def titles(n):
        t = rng.integers(1, V, size=(n, T))
        lens = rng.integers(1, T + 1, size=n)
        t[np.arange(T)[None, :] >= lens[:, None]] = 0
        return t

    hist = titles(B * W).reshape(B, W, T)
    npad = rng.integers(0, W, size=B)
    npad[rng.random(B) > pad_hist_frac] = 0
    for b in range(B):
        hist[b, :npad[b]] = 0
    cands = titles(B * K1).reshape(B, K1, T)
    user = rng.integers(0, N_u, size=B)
    return user.astype(np.int32), hist.astype(np.int32), cands.astype(np.int32)
