"""The C++ loaders and the generalized C++ feed (liblstur_data.so, include/lstur_data.h) against the Python mirror of the
reference's data path, which tests/test_data_path.py pins bit-exactly to vectors captured from the reference itself
(tests/golden/): DocMeta / ClickData / Vocab parsing incl. the hand-written edge set, the failures the reference raises,
validation and test generators, the pointwise and the `...Days` task variants."""
import os
import random
import shutil

import numpy as np
import pytest

from mnexp_amd import data_native, task, utils
from test_data_path import CASES, GOLD, G, make_config


@pytest.mark.parametrize('name', ['tiny', 'edge'])
def test_loaders_equal_golden_and_python(name):
    kw, dim, *_ = CASES[name]
    cfg = make_config(os.path.join(GOLD, name), **kw)
    h = task.get(cfg)                                            # the Python loaders (golden-pinned)
    table, present = data_native.load_docs(cfg.doc_meta_input, cfg.title_shape)
    assert table.dtype == np.int32 and np.array_equal(table, h.docs.title_table)
    assert np.array_equal(present, G[name + '_present']) and np.array_equal(present, h.docs.present)
    assert np.array_equal(table.astype(np.float64)[present], G[name + '_titles'][present])
    emb = data_native.load_vocab(cfg.title_embedding_input, dim)
    assert emb.dtype == np.float32 and np.array_equal(emb, G[name + '_embedding'])
    clicks = data_native.ClickData(cfg.training_data_input, present=present)
    assert len(clicks) == len(h.data)
    lazy = data_native.LazyUsers(clicks, type(h).Impression)
    for u, (tr, va) in enumerate(h.data):
        ltr, lva = lazy[u]
        for a, b in zip(tr + va, ltr + lva):
            assert a.pos == b.pos and a.neg == b.neg
            assert hasattr(a, 'time') == hasattr(b, 'time') and (not hasattr(a, 'time') or a.time == b.time)
        assert len(tr) == len(ltr) and len(va) == len(lva)
    if name == 'edge':
        assert np.array_equal(np.array([[len(a), len(b)] for a, b in lazy]), G['edge_imp_counts'])
        assert not clicks.timed


def test_loader_failures_are_the_references(tmp_path):
    """where the reference's parsers raise, the native loaders fail with the file and line"""
    def docmeta(lines):
        p = tmp_path / 'DocMeta.tsv'
        p.write_text(''.join(lines))
        return str(p)
    with pytest.raises(ValueError, match='line 2'):               # int('') on an empty token (double space)
        data_native.load_docs(docmeta(['a\t1\tn\tn\t1 2\t3\n', 'b\t2\tn\tn\t1  2\t3\n']), 4)
    with pytest.raises(ValueError, match='line 1'):               # an empty sentence: "#N#1 2" -> int('')
        data_native.load_docs(docmeta(['a\t1\tn\tn\t#N#1 2\t3\n']), 4)
    with pytest.raises(ValueError, match='columns'):              # IndexError in the reference
        data_native.load_docs(docmeta(['a\t1\tn\n']), 4)
    t, p = data_native.load_docs(docmeta(['a\t3\tn\tn\t7 8 9 10 11\tx\n', 'b\t3\tn\tn\t5#N#6 6\tx\n', 'c\t1\tn\tn\t2\tx']), 4)
    assert t.tolist() == [[0, 0, 0, 0], [2, 0, 0, 0], [0, 0, 0, 0], [5, 0, 0, 0]] and p.tolist() == [True, True, False, True]
    d = tmp_path / 'bad'
    shutil.copytree(os.path.join(GOLD, 'edge'), d)
    with open(d / 'ClickData.tsv', 'a') as f:
        f.write('u9\tt\t1#TAB#4\t\n')                             # doc 4 is a hole in DocMeta
    cfg = make_config(str(d), native_load=True, **CASES['edge'][0])
    with pytest.raises(ValueError, match='unknown doc id 4'):     # KeyError in the Python path (test_data_path.py)
        task.get(cfg)
    c = tmp_path / 'ClickData.tsv'
    c.write_text('u\tt\t1 2#TAB#\t\n')                            # empty negatives: int('') in the reference
    with pytest.raises(ValueError, match='invalid literal'):
        data_native.ClickData(str(c))
    c.write_text('u\tt\t1#TAB#2#TAB#13/01/2019 1:00:00 PM\t\n')   # strptime failure
    with pytest.raises(ValueError, match='time data'):
        data_native.ClickData(str(c))
    c.write_text('u\tt\t#TAB#9#N#1#TAB#2#TAB#12/31/2018 12:00:05 AM#TAB#x\t3#TAB#4#TAB#01/01/2019 12:00:00 PM\n\tq\t\t\n')
    k = data_native.ClickData(str(c))
    assert len(k) == 2 and k.timed
    assert k.train['imp_off'].tolist() == [0, 1, 1] and k.valid['imp_off'].tolist() == [0, 1, 1]
    from datetime import datetime
    assert k.train['time'][0] == (datetime(2018, 12, 31, 0, 0, 5) - datetime(1970, 1, 1)).total_seconds()
    assert k.valid['time'][0] == (datetime(2019, 1, 1, 12) - datetime(1970, 1, 1)).total_seconds()
    v = tmp_path / 'Vocab.tsv'
    v.write_text('w\t1\t1 2\n')
    with pytest.raises(ValueError, match='no row'):               # max() of an empty dict in the reference
        data_native.load_vocab(str(v), 3)


def _same_batch(a, b):
    (ia, ya), (ib, yb) = a, b
    assert len(ia) == len(ib)
    for x, y in zip(ia, ib):
        assert x.dtype == y.dtype and np.array_equal(x, y), (x, y)
    assert np.array_equal(ya, yb)


@pytest.mark.parametrize('tname,kw', [
    ('Seq2VecPaperSoftmaxId', {}), ('Seq2VecPaperSoftmax', {}),
    ('Seq2VecPaperSoftmaxDaysId', dict(days=0.05)), ('Seq2VecPaperSoftmaxDays', dict(days=2)),
    ('Seq2VecPaperId', {}), ('Seq2VecPaper', dict(arch='gru')), ('Seq2VecPaperDot', dict(arch='gru')),
])
@pytest.mark.parametrize('name', ['tiny', 'edge'])
def test_native_task_and_feed_equal_the_python_generators(name, tname, kw):
    """a natively loaded task (its Python generators on LazyUsers) and the C++ feed over the same ClickData, against the
    Python-loaded task: training samples, pool batches + RNG state, validation batches, every test impression"""
    if name == 'edge' and 'Days' in tname:
        pytest.skip('the edge set has impressions without a time stamp')
    base = dict(CASES[name][0], task=tname, **kw)
    seed = CASES[name][4]
    hp = task.get(make_config(os.path.join(GOLD, name), **base))
    hn = task.get(make_config(os.path.join(GOLD, name), native_load=True, native_feed=True, **base))
    assert hn.clicks is not None and hp.clicks is None
    assert np.array_equal(hn.docs.title_table, hp.docs.title_table)
    # the task's own Python generators on the natively loaded data
    for gen_name in ('train_gen_ids', 'valid_gen_ids'):
        np.random.seed(seed); random.seed(seed)
        a = [next(g) for g in [getattr(hp, gen_name)()] for _ in range(60)]
        np.random.seed(seed); random.seed(seed)
        b = [next(g) for g in [getattr(hn, gen_name)()] for _ in range(60)]
        _same_batch(hp._stack_ids(a), hn._stack_ids(b))
    # the C++ feed: pool batches (+ the generator state behind them), validation batches, test impressions
    feed = hn.native_feed()
    assert feed is not None
    feed.seed(seed + 1)
    np.random.seed(seed + 1); random.seed(seed + 1)
    it = hp.train_ids
    for _ in range(4):
        _same_batch(next(feed), next(it))
    assert np.array_equal(feed.peek_rng(), np.random.randint(0, 1 << 30, size=4))
    feed.seed(seed + 2)
    np.random.seed(seed + 2)
    itn, itp = hn.valid_ids, hp.valid_ids
    for _ in range(5):
        _same_batch(next(itn), next(itp))
    got = list(feed.test_impressions())
    want = [hp._stack_ids(rows) for rows in hp.test_gen_ids()]
    assert len(got) == len(want) > 0
    for a, b in zip(got, want):
        _same_batch(a, b)
    # dealt out over two ranks: every impression exactly once, in order
    parts = [list(feed.test_impressions(r, 2)) for r in range(2)]
    assert sorted(i for p in parts for i, _ in p) == list(range(len(want)))
    for p in parts:
        for i, b in p:
            _same_batch(b, want[i])


def test_pointwise_subsampling_matches_the_reference_fixture(tmp_path):
    """users with more than --max-impression training impressions train on random.sample(impressions, max_impression)
    (task/paper.py:271-272, Python's `random` module): the Python generators AND the C++ feed -- CPython's sample() restated
    on the MT19937 state the host hands over (include/lstur_data.h: lstur_data_pyrandom) -- reproduce the rows the reference
    itself generated (tests/golden/make_golden.py: max_impression = 3, both generators seeded with 31)"""
    cfg = make_config(os.path.join(GOLD, 'tiny'), task='Seq2VecPaperId', max_impression=3, native_load=True, native_feed=True)
    h = task.get(cfg)
    np.random.seed(31); random.seed(31)
    gen = h.train_gen_ids()
    X, Y = h._expand([next(gen) for _ in range(150)])
    assert np.array_equal(X[0], G['Seq2VecPaperId_train_user']) and np.array_equal(X[1], G['Seq2VecPaperId_train_clicked'])
    feed = h.native_feed()
    assert feed is not None
    feed.seed(31)
    user, hist, cands = feed.samples(150)
    np.random.seed(31); random.seed(31)
    gen = h.train_gen_ids()
    rows = [next(gen) for _ in range(150)]
    assert np.array_equal(user, np.array([r[0] for r in rows]))
    assert np.array_equal(hist, np.array([r[1] for r in rows]))
    assert np.array_equal(cands[:, 0], np.array([r[2][0] for r in rows]))
    assert np.array_equal(user, G['Seq2VecPaperId_train_user'].reshape(-1))
    # ... and the other branch of sample(): more impressions than its pool threshold (n = 30 > 21 with k = 3 <= 5) makes it
    # draw indices into a set, rejecting repeats; k = 7 of 30 stays on the pool branch (threshold 21 + 4^3)
    rng = np.random.RandomState(5)
    with open(tmp_path / 'DocMeta.tsv', 'w') as f:
        for i in range(1, 61):
            f.write('d%d\t%d\tnews\tus\t%s\t1 2\n' % (i, i, ' '.join(str(x) for x in rng.randint(1, 40, size=5))))
    with open(tmp_path / 'ClickData.tsv', 'w') as f:
        for u in range(4):
            imps = ['%d#TAB#%d %d#TAB#01/02/2019 03:04:05 PM#TAB#x' % tuple(rng.randint(1, 61, size=3)) for _ in range(30)]
            f.write('u%d\tt\t%s\t%s\n' % (u, '#N#'.join(imps), imps[0]))
    open(tmp_path / 'Vocab.tsv', 'w').close()
    for k in (3, 7):
        cfg2 = make_config(str(tmp_path), task='Seq2VecPaperId', max_impression=k, native_load=True, native_feed=True,
                           title_shape=5, window_size=4)
        h2 = task.get(cfg2)
        f2 = h2.native_feed()
        f2.seed(7)
        u2, hi2, c2 = f2.samples(300)
        np.random.seed(7); random.seed(7)
        g2 = h2.train_gen_ids()
        r2 = [next(g2) for _ in range(300)]
        assert np.array_equal(u2, np.array([r[0] for r in r2])) and np.array_equal(c2[:, 0], np.array([r[2][0] for r in r2]))
        assert np.array_equal(hi2, np.array([r[1] for r in r2]))
        assert len(set(u2.tolist())) == 4


def test_native_load_is_fast_at_scale(tmp_path):
    """10 k users / 5 k docs load in well under a second (1 % of BASELINE.json configs[3]; the synthetic files themselves are
    written by Python loops, which is what bounds this test); bench.py's e2e leg prints the load time at 1 M users / 200 k docs"""
    import time
    from mnexp_amd import synth
    rng = np.random.default_rng(0)
    topic, titles = synth.make_docs(5000, 50000, 30, rng)
    synth.write_docmeta(str(tmp_path / 'DocMeta.tsv'), titles)
    synth.write_clickdata(str(tmp_path / 'ClickData.tsv'), 10000, topic, 20, 4, rng)
    t0 = time.time()
    table, present = data_native.load_docs(str(tmp_path / 'DocMeta.tsv'), 30)
    clicks = data_native.ClickData(str(tmp_path / 'ClickData.tsv'), present=present)
    dt = time.time() - t0
    assert len(clicks) == 10000 and clicks.train['imp_off'][-1] > 150000
    assert dt < 3.0, dt
