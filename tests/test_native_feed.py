"""The C++ sample feed (liblstur_data.so) against the Python task generators, which are themselves pinned
bit-exactly to the reference's data path by tests/test_data_path.py + tests/golden/."""
import os
import random
import re

import numpy as np
import pytest

from mnexp_amd import data_native, task
from test_data_path import CASES, GOLD, G, make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    text = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'lstur_data.h')).read(), flags=re.S)
    syms = sorted(set(re.findall(r'\b(lstur_(?:data|docs|clicks|vocab|synth)_[a-z0-9_]+)\s*\(', text)))
    assert syms == sorted(data_native.SIGNATURES)
    for s in syms:
        assert hasattr(data_native.lib, s)


@pytest.mark.parametrize('name', ['tiny', 'edge'])
def test_batches_equal_python_path_and_golden(name):
    kw, dim, ns, nb, seed = CASES[name]
    cfg = make_config(os.path.join(GOLD, name), **kw)
    h = task.get(cfg)
    feed = data_native.NativeTrainFeed(h, rank=0, world=1)
    # raw generator samples
    feed.seed(seed)
    user, hist, cands = feed.samples(ns)
    np.random.seed(seed)
    gen = h.train_gen_ids()
    py = [next(gen) for _ in range(ns)]
    assert np.array_equal(user, [s[0] for s in py])
    assert np.array_equal(hist, np.stack([s[1] for s in py]))
    assert np.array_equal(cands, np.array([s[2] for s in py]))
    # pool-shuffled batches: against the python path AND the reference's golden batches
    feed.seed(seed + 1)
    np.random.seed(seed + 1)
    it = h.train_ids
    for b in range(nb):
        (u1, h1, c1), y1 = next(feed)
        (u2, h2, c2), y2 = next(it)
        assert np.array_equal(u1, u2) and np.array_equal(h1, h2) and np.array_equal(c1, c2) and np.array_equal(y1, y2)
        assert u1.dtype == u2.dtype and h1.dtype == h2.dtype and c1.dtype == c2.dtype
        assert np.array_equal(h.docs.titles_f64[h1], G['%s_batch%d_clicked' % (name, b)])
        assert np.array_equal(h.docs.titles_f64[c1], G['%s_batch%d_cands' % (name, b)])
        assert np.array_equal(u1, G['%s_batch%d_user' % (name, b)])
    # the generator state after those batches equals NumPy's
    assert np.array_equal(feed.peek_rng(), np.random.randint(0, 1 << 30, size=4))
    assert np.array_equal(feed.peek_rng(), G[name + '_rng_after_batches'])


def test_sharded_feed_matches_sharded_python(monkeypatch):
    from mnexp_amd import dist as D
    cfg = make_config(os.path.join(GOLD, 'tiny'))
    h = task.get(cfg)
    for r in range(2):
        monkeypatch.setattr(D, 'world_size', lambda group=None: 2)
        monkeypatch.setattr(D, 'rank', lambda group=None, r=r: r)
        feed = data_native.NativeTrainFeed(h, seed=3)
        np.random.seed(3)
        it = h.train_ids
        for _ in range(3):
            (u1, h1, c1), _y = next(feed)
            (u2, h2, c2), _y2 = next(it)
            assert np.array_equal(u1, u2) and np.array_equal(h1, h2) and np.array_equal(c1, c2)
            assert (u1 % 2 == r).all()


def test_feed_is_fast():
    import time
    cfg = make_config(os.path.join(GOLD, 'tiny'), batch_size=64, window_size=50)
    h = task.get(cfg)
    feed = data_native.NativeTrainFeed(h, seed=0, rank=0, world=1)
    next(feed)
    t = time.time()
    for _ in range(200):
        next(feed)
    rate = 200 * 64 / (time.time() - t)
    assert rate > 100000, rate
