"""Host data path vs golden vectors captured from the reference's own Python data path
(tests/golden/make_golden.py; bit-exact: parsing, windows, negative sampling, pool shuffle)."""
import json
import os
import random

import numpy as np
import pytest

from mnexp_amd import document, settings, task, utils

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
G = np.load(os.path.join(GOLD, 'data_path.npz'))


def make_config(data_dir, **kw):
    settings.public_config = dict(input_training_data_path=data_dir, input_validation_data_path=data_dir,
                                  input_previous_model_path='/tmp/_m', output_model_path='/tmp/_m', log_dir='/tmp/_l')
    base = dict(task='Seq2VecPaperSoftmaxId', arch='igru', batch_size=8, window_size=5, title_shape=8,
                body_shape=4, negative_samples=4, training_step=20, validation_step=2, max_impression=200,
                news_encoder='cnnatt', textual_embedding_dim=16, days=30, debug=False, epochs=2,
                validation_impression=10, testing_impression=10, learning_rate=1e-3, learning_rate_decay=0.2,
                dropout=0.0, user_embedding_dim=8, title_filter_shape=(16, 3), score_model='dot',
                enable_pretrain_encoder=False, textual_embedding_trainable=False, name='')
    base.update(kw)
    return settings.Config(base, verbose=False)


CASES = {
    'tiny': (dict(), 16, 200, 10, 11),
    'edge': (dict(window_size=3, title_shape=6, batch_size=2, negative_samples=2, textual_embedding_dim=4), 4, 30, 4, 5),
}


@pytest.fixture(scope='module', params=['tiny', 'edge'])
def case(request):
    name = request.param
    kw, dim, ns, nb, seed = CASES[name]
    cfg = make_config(os.path.join(GOLD, name), **kw)
    return name, cfg, task.get(cfg), dim, ns, nb, seed


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    np.testing.assert_array_equal(a, b)


def test_docs_and_embedding(case):
    name, cfg, h, dim, *_ = case
    present = G[name + '_present']
    assert h.doc_count == len(present)
    same(h.docs.present, present)
    titles = G[name + '_titles']
    for i in np.flatnonzero(present):
        same(h.docs[int(i)].title, titles[i])
    for i in np.flatnonzero(~present):
        with pytest.raises(KeyError):
            h.docs[int(i)]
    same(utils.load_textual_embedding(cfg.title_embedding_input, dim), G[name + '_embedding'])


def test_train_samples(case):
    name, cfg, h, dim, ns, nb, seed = case
    np.random.seed(seed)
    random.seed(seed)
    gen = h.train_gen_ids()
    samples = [next(gen) for _ in range(ns)]
    X, Y = h._expand(samples)
    same(X[0], G[name + '_train_user'])
    same(X[1], G[name + '_train_clicked'])
    same(np.stack(X[2:], axis=1), G[name + '_train_cands'])
    same(Y, G[name + '_train_label'])


def test_pool_batches_and_rng_stream(case):
    name, cfg, h, dim, ns, nb, seed = case
    np.random.seed(seed + 1)
    random.seed(seed + 1)
    it = h.train
    for b in range(nb):
        X, Y = next(it)
        same(X[0], G['%s_batch%d_user' % (name, b)])
        same(X[1], G['%s_batch%d_clicked' % (name, b)])
        same(np.stack(X[2:], axis=1), G['%s_batch%d_cands' % (name, b)])
        same(Y, G['%s_batch%d_y' % (name, b)])
    same(np.random.randint(0, 1 << 30, size=4), G[name + '_rng_after_batches'])


def test_id_form_batches_equal_expanded(case):
    name, cfg, h, dim, ns, nb, seed = case
    np.random.seed(seed + 1)
    it = h.train_ids
    (user, hist, cands), Y = next(it)
    same(user, G['%s_batch0_user' % name])
    same(h.docs.titles_f64[hist], G['%s_batch0_clicked' % name])
    same(h.docs.titles_f64[cands], G['%s_batch0_cands' % name])
    assert hist.dtype == np.int32 and cands.dtype == np.int32


def test_valid_batches(case):
    name, cfg, h, dim, ns, nb, seed = case
    np.random.seed(seed + 2)
    it = h.valid
    for b in range(3):
        X, Y = next(it)
        same(X[0], G['%s_valid%d_user' % (name, b)])
        same(X[1], G['%s_valid%d_clicked' % (name, b)])
        same(np.stack(X[2:], axis=1), G['%s_valid%d_cands' % (name, b)])
        same(Y, G['%s_valid%d_y' % (name, b)])


def test_test_impressions(case):
    name, cfg, h, dim, ns, nb, seed = case
    n = int(G[name + '_n_test'])
    got = 0
    for i, rows in zip(range(n), h.test_gen_ids()):
        X, labels = h._expand(rows)
        same(X[0], G['%s_test%d_user' % (name, i)])
        same(X[1], G['%s_test%d_clicked' % (name, i)])
        same(X[2], G['%s_test%d_cand' % (name, i)])
        same(labels, G['%s_test%d_label' % (name, i)])
        got += 1
    assert got == n


def test_edge_impression_counts():
    cfg = make_config(os.path.join(GOLD, 'edge'), **CASES['edge'][0])
    h = task.get(cfg)
    same(np.array([[len(a), len(b)] for a, b in h.data]), G['edge_imp_counts'])
    assert hasattr(h.data[0][0][0], 'time') and not hasattr(h.data[3][0][0], 'time')


def test_parsers_and_ranking_metrics():
    with open(os.path.join(GOLD, 'metrics.json')) as f:
        gold = json.load(f)
    pd = document.DocumentParser(document.parse_document(), document.pad_document(2, 4))
    for text, want in gold['parsed'].items():
        got = pd(text)
        assert got.dtype == np.float64 and got.tolist() == want
    with pytest.raises(ValueError):
        pd('#N#1 2')
    for c in gold['ranking']:
        y, s = np.array(c['y_true']), np.array(c['y_score'])
        assert utils.dcg_score(y, s, 10) == c['dcg10']
        assert utils.ndcg_score(y, s, 10) == c['ndcg10']
        assert utils.ndcg_score(y, s, 5) == c['ndcg5']
        assert utils.mrr_score(y, s) == c['mrr']


def test_unknown_doc_id_fails_at_load(tmp_path):
    import shutil
    d = tmp_path / 'bad'
    shutil.copytree(os.path.join(GOLD, 'edge'), d)
    with open(d / 'ClickData.tsv', 'a') as f:
        f.write('u9\tt\t1#TAB#4\t\n')          # doc 4 is a hole in DocMeta
    cfg = make_config(str(d), **CASES['edge'][0])
    with pytest.raises(KeyError):
        task.get(cfg)


@pytest.mark.parametrize('tname,has_user', [('Seq2VecPaperSoftmaxDaysId', True), ('Seq2VecPaperSoftmaxDays', False)])
def test_time_window_variants(tname, has_user):
    """History entries expire `days` after the click and read as doc 0 in place (paper.py:668-881)."""
    cfg = make_config(os.path.join(GOLD, 'tiny'), task=tname, days=0.05)
    h = task.get(cfg)
    u = 1 if has_user else 0
    np.random.seed(21)
    random.seed(21)
    gen = h.train_gen_ids()
    X, Y = h._expand([next(gen) for _ in range(120)])
    if has_user:
        same(X[0], G[tname + '_train_user'])
    same(X[u], G[tname + '_train_clicked'])
    same(np.stack(X[u + 1:], axis=1), G[tname + '_train_cands'])
    same(Y, G[tname + '_train_label'])
    # expiry really happened: with a huge window the same walk yields more samples / fuller histories
    h_inf = task.get(make_config(os.path.join(GOLD, 'tiny'), task=tname, days=10000))
    np.random.seed(21)
    gen = h_inf.train_gen_ids()
    X_inf, _ = h_inf._expand([next(gen) for _ in range(120)])
    assert not np.array_equal(X_inf[u], X[u])
    np.random.seed(22)
    random.seed(22)
    it = h.train
    for b in range(4):
        X, Y = next(it)
        if has_user:
            same(X[0], G['%s_batch%d_user' % (tname, b)])
        same(X[u], G['%s_batch%d_clicked' % (tname, b)])
        same(np.stack(X[u + 1:], axis=1), G['%s_batch%d_cands' % (tname, b)])
    same(np.random.randint(0, 1 << 30, size=4), G[tname + '_rng_after_batches'])
    np.random.seed(23)
    it = h.valid
    for b in range(3):
        X, Y = next(it)
        same(X[u], G['%s_valid%d_clicked' % (tname, b)])
        same(np.stack(X[u + 1:], axis=1), G['%s_valid%d_cands' % (tname, b)])
    for i, rows in zip(range(int(G[tname + '_n_test'])), h.test_gen_ids()):
        X, labels = h._expand(rows)
        same(X[u], G['%s_test%d_clicked' % (tname, i)])
        same(X[u + 1], G['%s_test%d_cand' % (tname, i)])
        same(labels, G['%s_test%d_label' % (tname, i)])


@pytest.mark.parametrize('tname,has_user,arch', [('Seq2VecPaperId', True, 'igru'), ('Seq2VecPaper', False, 'gru')])
def test_pointwise_sibling_data_path(tname, has_user, arch):
    """one candidate per row (the positive, then K sampled negatives), ``random.sample`` of
    ``max_impression`` training impressions per user (paper.py:38-97, 266-328) -- bit-exact against the reference,
    including how both global RNG streams (``random`` and ``np.random``) are consumed."""
    cfg = make_config(os.path.join(GOLD, 'tiny'), task=tname, arch=arch, max_impression=3)
    h = task.get(cfg)
    u = 1 if has_user else 0
    np.random.seed(31)
    random.seed(31)
    gen = h.train_gen_ids()
    X, Y = h._expand([next(gen) for _ in range(150)])
    if has_user:
        same(X[0], G[tname + '_train_user'])
    same(X[u], G[tname + '_train_clicked'])
    same(np.stack(X[u + 1:], axis=1), G[tname + '_train_cands'])
    same(Y, G[tname + '_train_label'])
    assert len(X) == u + 2 and set(np.unique(Y)) == {0, 1}
    # the sub-sampling really happened: without the cap the same walk yields different rows
    h_all = task.get(make_config(os.path.join(GOLD, 'tiny'), task=tname, arch=arch, max_impression=200))
    np.random.seed(31)
    random.seed(31)
    gen = h_all.train_gen_ids()
    X_all, _ = h_all._expand([next(gen) for _ in range(150)])
    assert not np.array_equal(X_all[u], X[u])
    np.random.seed(32)
    random.seed(32)
    it = h.train
    for b in range(4):
        X, Y = next(it)
        if has_user:
            same(X[0], G['%s_batch%d_user' % (tname, b)])
        same(X[u], G['%s_batch%d_clicked' % (tname, b)])
        same(np.stack(X[u + 1:], axis=1), G['%s_batch%d_cands' % (tname, b)])
        same(Y, G['%s_batch%d_y' % (tname, b)])
    same(np.random.randint(0, 1 << 30, size=4), G[tname + '_rng_after_batches'])
    np.random.seed(33)
    it = h.valid
    for b in range(3):
        X, Y = next(it)
        same(X[u], G['%s_valid%d_clicked' % (tname, b)])
        same(np.stack(X[u + 1:], axis=1), G['%s_valid%d_cands' % (tname, b)])
        same(Y, G['%s_valid%d_y' % (tname, b)])
    for i, rows in zip(range(int(G[tname + '_n_test'])), h.test_gen_ids()):
        X, labels = h._expand(rows)
        same(X[u], G['%s_test%d_clicked' % (tname, i)])
        same(X[u + 1], G['%s_test%d_cand' % (tname, i)])
        same(labels, G['%s_test%d_label' % (tname, i)])
