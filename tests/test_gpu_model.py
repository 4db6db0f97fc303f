"""GPU parity of the whole native LSTUR step against the CPU oracle (through the C-ABI).

Tolerance: fp32 logits within 1e-3 relative (BASELINE.json north_star); gradients and updated
parameters are held to 2e-3 of the tensor's max-abs plus a small absolute floor (fp32 kernels vs the
fp64 oracle).  Index / mask work (dropout masks, padding masks) must match exactly, which is checked
through values that would be visibly wrong otherwise (dropout on, ragged titles, padded histories).
"""
import numpy as np
import pytest

from conftest import make_batch

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')


def rel_close(got, ref, tol=2e-3, name=''):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = np.abs(ref).max() + 1e-12
    err = np.abs(got - ref).max()
    assert err <= tol * scale + 1e-7, '{}: max err {:.3e} vs scale {:.3e}'.format(name, err, scale)


def build(arch, dims, B, W, T, K1, drop, seed=5, **model_kw):
    from oracle import lstur_oracle as O
    from mnexp_amd.native_model import NativeModel
    rng = np.random.default_rng(seed)
    p = O.init_params(rng, arch=arch, dtype=np.float64, **dims)
    for n in ('conv_b', 'att_b', 'dense_b', 'gru_b'):
        p[n] = rng.normal(0, 0.1, size=p[n].shape)
    model = NativeModel(p['word_emb'], dims['N_u'], B, W, T, K1 - 1, dims['F'], dims['k'], dims['U'], arch=arch,
                        dropout=drop, lr=1e-3, seed=77, **model_kw)
    model.set_weights({n: p[n] for n in p if n != 'word_emb'})
    return O, p, model, rng


TINY = dict(V=60, D=12, F=16, k=3, U=8, N_u=11)
MID = dict(V=500, D=48, F=80, k=3, U=40, N_u=64)
ODD = dict(V=90, D=20, F=23, k=5, U=18, N_u=9)      # F, U not multiples of 16; k = 5


@pytest.mark.parametrize('arch', ['igru', 'gru', 'hgru', 'nigru'])
@pytest.mark.parametrize('dims,B,W,T,K1,drop', [
    (TINY, 5, 4, 7, 3, 0.0),
    (TINY, 5, 4, 7, 3, 0.25),
    (MID, 19, 9, 11, 5, 0.2),
    (ODD, 7, 3, 6, 2, 0.0),
    (TINY, 4, 3, 33, 2, 0.25),     # T > 32: the streaming attention kernels (title-resident path covers T <= 32)
    (TINY, 4, 3, 32, 2, 0.25),
])
def test_forward_backward_matches_oracle(arch, dims, B, W, T, K1, drop):
    _need_gpu()
    O, p, model, rng = build(arch, dims, B, W, T, K1, drop)
    user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
    user[1] = user[0]
    hist[2] = 0                                    # a sample whose whole history is padding
    model.step = 3
    ref = O.forward_backward(p, arch, user, hist, cands, drop_p=drop, seed=77, step=3)
    import ctypes as C
    from mnexp_amd._lib import lib, check
    u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 3, 1.0,
                                     model.stats.data_ptr(), model._stream()), 'fb')
    torch.cuda.synchronize()
    loss, acc = model.stats.tolist()
    logits = model.peek('logits').cpu().numpy().reshape(B, K1)
    np.testing.assert_allclose(logits, ref['logits'], rtol=1e-3, atol=1e-5)
    assert abs(loss - ref['loss']) <= 1e-4 * max(1.0, abs(ref['loss']))
    assert abs(acc - ref['acc']) < 1e-6
    ref_doc = ref['doc_vec'].copy()                # the library masks padded history slots in place
    ref_doc[:B * W] *= (hist != 0).any(-1).reshape(-1, 1)
    got_doc = model.peek('doc_vec').cpu().numpy().reshape(-1, dims['U'])
    rel_close(got_doc, ref_doc, 1e-3, 'doc_vec')
    g = model.get_grads()
    for n in O.dense_param_names(arch):
        rel_close(g[n], ref['grads'][n], 2e-3, n)
    assert np.array_equal(g['user_rows'], user)
    if arch != 'nigru':
        rel_close(g['user_row_grads'], ref['user_row_grads'], 2e-3, 'user_row_grads')


@pytest.mark.parametrize('arch', ['igru', 'gru', 'nigru'])
def test_three_adam_steps_track_oracle(arch):
    _need_gpu()
    dims, B, W, T, K1 = MID, 16, 6, 9, 5
    O, p, model, rng = build(arch, dims, B, W, T, K1, 0.2)
    p32 = {n: a.astype(np.float32) for n, a in p.items()}
    names = O.dense_param_names(arch) + ['user_emb']
    opt = O.Adam(p32, names, lr=1e-3)
    for step in range(3):
        user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
        user[3] = user[0]
        O.train_step(p32, arch, opt, user, hist, cands, drop_p=0.2, seed=77, step=step)
        X = [user.astype(np.int64), hist.astype(np.float64)] + [cands[:, j].astype(np.float64) for j in range(K1)]
        model.train_on_batch(X, None)
    torch.cuda.synchronize()
    w = model.get_weights()
    for n in names:
        # after 3 Adam steps every touched weight moved by ~3e-3; the update must agree to a few %
        moved = np.abs(p32[n].astype(np.float64) - p[n]).max()
        err = np.abs(w[n].astype(np.float64) - p32[n]).max()
        assert err <= 0.05 * moved + 2e-6, (n, err, moved)


@pytest.mark.parametrize('dims,B,W,T,K1,drop', [
    (TINY, 5, 4, 7, 3, 0.0),
    (TINY, 5, 4, 7, 3, 0.25),
    (MID, 19, 9, 11, 5, 0.2),
    (dict(V=90, D=20, F=24, k=5, U=18, N_u=9), 7, 3, 6, 2, 0.0),   # k = 5, D = 20: ragged column tile of the flipped kernel
    (dict(V=400, D=300, F=400, k=3, U=32, N_u=7), 3, 4, 30, 2, 0.2),   # BASELINE D / F / T: the 19-tile configuration
])
def test_word_table_gradient_matches_oracle(dims, B, W, T, K1, drop):
    """--textual-embedding-trainable (paper.py:132-138, main.py:36): the conv's input gradient, through the embedding
    dropout mask, scatter-added into the word table's gradient — the pad row 0 included (a pad token's embedding is
    an input of its neighbours' windows even though its own conv output is masked)."""
    _need_gpu()
    O, p, model, rng = build('igru', dims, B, W, T, K1, drop, word_trainable=True)
    p['word_emb'][0] = rng.normal(0, 0.1, size=dims['D'])      # a trained table's pad row is no longer zero
    model.set_weights({'word_emb': p['word_emb']})
    user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
    hist[1] = 0
    model.step = 2
    ref = O.forward_backward(p, 'igru', user, hist, cands, drop_p=drop, seed=77, step=2, word_trainable=True)
    from mnexp_amd._lib import lib, check
    u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 2, 1.0,
                                     model.stats.data_ptr(), model._stream()), 'fb')
    torch.cuda.synchronize()
    logits = model.peek('logits').cpu().numpy().reshape(B, K1)
    np.testing.assert_allclose(logits, ref['logits'], rtol=1e-3, atol=1e-5)
    g = model.get_grads()
    for n in O.dense_param_names('igru'):
        rel_close(g[n], ref['grads'][n], 2e-3, n)
    gw, rw = g['word_emb'], ref['grads']['word_emb']
    rel_close(gw, rw, 2e-3, 'word_emb')
    assert np.abs(rw[0]).max() > 0                              # the pad row does receive gradient
    untouched = np.setdiff1d(np.arange(dims['V']), np.concatenate([hist.reshape(-1), cands.reshape(-1)]))
    assert not gw[untouched].any()                              # rows of absent words stay exactly 0


def test_word_table_adam_steps_track_oracle():
    """TF-1 Keras Adam densifies the IndexedSlices gradient of the word table: every row's moments decay each step."""
    _need_gpu()
    dims, B, W, T, K1 = MID, 16, 6, 9, 5
    O, p, model, rng = build('igru', dims, B, W, T, K1, 0.2, word_trainable=True)
    p32 = {n: a.astype(np.float32) for n, a in p.items()}
    names = O.dense_param_names('igru') + ['user_emb', 'word_emb']
    opt = O.Adam(p32, names, lr=1e-3)
    for step in range(3):
        user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
        O.train_step(p32, 'igru', opt, user, hist, cands, drop_p=0.2, seed=77, step=step, word_trainable=True)
        X = [user.astype(np.int64), hist.astype(np.float64)] + [cands[:, j].astype(np.float64) for j in range(K1)]
        model.train_on_batch(X, None)
    torch.cuda.synchronize()
    w = model.get_weights()
    w['word_emb'] = model.word_emb.cpu().numpy()
    for n in names:
        moved = np.abs(p32[n].astype(np.float64) - p[n]).max()
        err = np.abs(w[n].astype(np.float64) - p32[n]).max()
        assert moved > 1e-4, n
        assert err <= 0.05 * moved + 2e-6, (n, err, moved)


@pytest.mark.parametrize('arch,score_model', [('igru', 'ddot'), ('igru', 'dnn'), ('gru', 'ddot'), ('hgru', 'dnn'),
                                              ('ngru', 'ddot'), ('ngru', 'dnn'), ('dgru', 'ddot'), ('dgru', 'dnn'),
                                              ('pgru', 'dot'), ('vo', 'dot'), ('niavg', 'dot'), ('pgru', 'ddot'),
                                              ('niatt', 'dot'), ('niatt', 'dnn'), ('iigru', 'dot'), ('iigru', 'ddot')])
def test_score_models_and_wide_user_vectors_match_oracle(arch, score_model):
    """--score-model ddot | dnn (paper.py:448-455), and -a ngru | dgru (paper.py:600-602, 608-611: 2U-wide user vectors;
    dgru drops a sample's whole user embedding with p = 0.5 in training): logits, every gradient, three Adam steps and
    the test model sigmoid(score_model([u, d])).  Also the minor user encoders `pgru`, `vo`, `niavg` (paper.py:622-630)."""
    _need_gpu()
    from oracle import lstur_oracle as O
    from mnexp_amd.native_model import NativeModel
    dims, B, W, T, K1, drop = MID, 19, 6, 9, 5, 0.2
    rng = np.random.default_rng(21)
    p = O.init_params(rng, arch=arch, dtype=np.float64, score_model=score_model, **dims)
    for n in p:
        if n.endswith('_b'):
            p[n] = rng.normal(0, 0.1, size=p[n].shape)
    model = NativeModel(p['word_emb'], dims['N_u'], B, W, T, K1 - 1, dims['F'], dims['k'], dims['U'], arch=arch,
                        dropout=drop, lr=1e-3, seed=77, score_model=score_model)
    model.set_weights({n: p[n] for n in p if n != 'word_emb'})
    names = O.dense_param_names(arch, score_model=score_model)
    tables = ['user_emb'] + (['user_emb2'] if arch == 'iigru' else [])     # iigru: a second user table (paper.py:617)
    assert set(names) | set(tables) == set(model.param_names())
    user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
    user[1] = user[0]
    hist[2] = 0
    model.step = 3
    ref = O.forward_backward(p, arch, user, hist, cands, drop_p=drop, seed=77, step=3)
    from mnexp_amd._lib import lib, check
    u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 3, 1.0,
                                     model.stats.data_ptr(), model._stream()), 'fb')
    torch.cuda.synchronize()
    loss, acc = model.stats.tolist()
    logits = model.peek('logits').cpu().numpy().reshape(B, K1)
    np.testing.assert_allclose(logits, ref['logits'], rtol=1e-3, atol=1e-5)
    assert abs(loss - ref['loss']) <= 1e-4 * max(1.0, abs(ref['loss'])) and abs(acc - ref['acc']) < 1e-6
    g = model.get_grads()
    for n in names:
        rel_close(g[n], ref['grads'][n], 2e-3, n)
    if arch not in ('niavg', 'niatt'):      # no user embedding in these graphs (paper.py:629-630, 207-209)
        rel_close(g['user_row_grads'], ref['user_row_grads'], 2e-3, 'user_row_grads')
    if arch == 'iigru':
        rel_close(model.peek('duser2').cpu().numpy().reshape(B, -1), ref['user_row_grads2'], 2e-3, 'user_row_grads2')
    if arch in ('vo', 'niavg', 'niatt'):
        assert not g['gru_w'].any() and not g['gru_u'].any() and not g['gru_b'].any()     # no GRU in these graphs
    if arch == 'dgru':       # dropped samples: exactly no gradient for their user row
        keep = O.dropout_keep(O.dropout_key(77, 3, 2), B, 0.5)
        assert 0 < keep.sum() < B and not g['user_row_grads'][~keep].any()
    # three Adam steps
    p32 = {n: a.astype(np.float32) for n, a in p.items()}
    opt = O.Adam(p32, names + tables, lr=1e-3)
    model.step = 0
    for step in range(3):
        user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
        O.train_step(p32, arch, opt, user, hist, cands, drop_p=drop, seed=77, step=step)
        X = [user.astype(np.int64), hist.astype(np.float64)] + [cands[:, j].astype(np.float64) for j in range(K1)]
        model.train_on_batch(X, None)
    torch.cuda.synchronize()
    w = model.get_weights()
    for n in names + tables:
        if n == 'sn2_b':     # softmax is shift-invariant: this gradient is identically 0, Adam only normalises rounding noise
            continue
        moved = np.abs(p32[n].astype(np.float64) - p[n]).max()
        err = np.abs(w[n].astype(np.float64) - p32[n]).max()
        assert err <= 0.05 * moved + 2e-6, (n, err, moved)
    # test model
    n = 23
    user, hist, cands = make_batch(rng, n, W, T, 1, dims['V'], dims['N_u'])
    got = model.predict([user.astype(np.int64), hist.astype(np.float64), cands[:, 0].astype(np.float64)])
    pw = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    pw['word_emb'] = p['word_emb']
    np.testing.assert_allclose(got[:, 0], O.predict(pw, arch, user, hist, cands[:, 0]), rtol=1e-3, atol=1e-6)


def test_all_pad_batch_and_short_batch():
    """Edge cases of the compacted token list: a batch whose every title is padding (count = 0: no conv work at all,
    exact-zero encoder gradients) and a batch shorter than max_batch."""
    _need_gpu()
    O, p, model, rng = build('igru', MID, 16, 6, 9, 5, 0.2)
    from mnexp_amd._lib import lib, check
    B, W, T, K1 = 16, 6, 9, 5
    user = rng.integers(0, MID['N_u'], size=B).astype(np.int32)
    hist = np.zeros((B, W, T), np.int32)
    cands = np.zeros((B, K1, T), np.int32)
    ref = O.forward_backward(p, 'igru', user, hist, cands, drop_p=0.2, seed=77, step=0)
    u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 0, 1.0,
                                     model.stats.data_ptr(), model._stream()), 'fb')
    torch.cuda.synchronize()
    loss, _ = model.stats.tolist()
    assert abs(loss - ref['loss']) < 1e-5 and np.isfinite(loss)
    g = model.get_grads()
    for n in ('conv_w', 'conv_b', 'att_w'):
        assert not g[n].any() and not ref['grads'][n].any(), n
    rel_close(g['dense_b'], ref['grads']['dense_b'], 2e-3, 'dense_b')
    # n < max_batch
    n = 5
    user, hist, cands = make_batch(rng, n, W, T, K1, MID['V'], MID['N_u'])
    ref = O.forward_backward(p, 'igru', user, hist, cands, drop_p=0.2, seed=77, step=1)
    u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), n, 1, 1.0,
                                     model.stats.data_ptr(), model._stream()), 'fb')
    torch.cuda.synchronize()
    g = model.get_grads()
    for nm in O.dense_param_names('igru'):
        rel_close(g[nm], ref['grads'][nm], 2e-3, nm)


def test_predict_and_evaluate_match_oracle():
    _need_gpu()
    dims, B, W, T, K1 = MID, 16, 6, 9, 5
    O, p, model, rng = build('igru', dims, B, W, T, K1, 0.2)
    n = 37                                           # > max_predict_rows -> chunked
    user, hist, cands = make_batch(rng, n, W, T, 1, dims['V'], dims['N_u'])
    X = [user.astype(np.int64), hist.astype(np.float64), cands[:, 0].astype(np.float64)]
    got = model.predict(X)
    ref = O.predict(p, 'igru', user, hist, cands[:, 0])
    assert got.shape == (n, 1)
    np.testing.assert_allclose(got[:, 0], ref, rtol=1e-3, atol=1e-6)
    model.set_title_table(np.arange(dims['V'] * T, dtype=np.int32).reshape(dims['V'], T) % dims['V'])
    tt = model.title_table.cpu().numpy()
    hid = rng.integers(0, dims['V'], size=(n, W)).astype(np.int32)
    cid = rng.integers(0, dims['V'], size=(n, 1)).astype(np.int32)
    got2 = model.predict_ids(user, hid, cid)
    ref2 = O.predict(p, 'igru', user, tt[hid], tt[cid[:, 0]])
    np.testing.assert_allclose(got2[:, 0], ref2, rtol=1e-3, atol=1e-6)
    # evaluate_generator: no dropout, mean loss / accuracy
    user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
    Xb = [user.astype(np.int64), hist.astype(np.float64)] + [cands[:, j].astype(np.float64) for j in range(K1)]
    loss, acc = model.evaluate_generator(iter([(Xb, None)]), 1)
    ref = O.forward_backward(p, 'igru', user, hist, cands, drop_p=0.0, want_grads=False)
    assert abs(loss - ref['loss']) < 1e-4 and abs(acc - ref['acc']) < 1e-6


def test_errors_are_reported_not_thrown_across_the_abi():
    _need_gpu()
    from mnexp_amd import _lib
    dims, B, W, T, K1 = TINY, 4, 3, 5, 2
    O, p, model, rng = build('igru', dims, B, W, T, K1, 0.0)
    user, hist, cands = make_batch(rng, B + 1, W, T, K1, dims['V'], dims['N_u'])
    X = [user.astype(np.int64), hist.astype(np.float64)] + [cands[:, j].astype(np.float64) for j in range(K1)]
    with pytest.raises(_lib.LsturError, match='max_batch'):
        model.train_on_batch(X, None)
    with pytest.raises(_lib.LsturError, match='multiple of 4'):
        _lib.query_sizes(_lib.make_cfg(4, 3, 5, 2, 10, 6, 8, 3, 8, 4))


@pytest.mark.parametrize('arch', ['igru', 'gru', 'hgru', 'iigru', 'vo'])
def test_id_keep_user_embedding_masking(arch):
    """--id-keep (the LSTUR paper's user-embedding masking; reference task/cook.py:139-142): logits, gradients and the row
    gradients of the user table(s) against the oracle with the same row mask; inference ignores it."""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    dims, B, W, T, K1 = MID, 16, 6, 9, 5
    O, p, model, rng = build(arch, dims, B, W, T, K1, 0.2, id_keep=0.7)
    user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
    ref = O.forward_backward(p, arch, user, hist, cands, drop_p=0.2, seed=77, step=0, id_keep=0.7)
    keep = O.dropout_keep(O.dropout_key(77, 0, 2), B, 1.0 - 0.7)
    assert 0 < keep.sum() < B
    u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 0, 1.0,
                                     model.stats.data_ptr(), model._stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(model.peek('logits').cpu().numpy().reshape(B, K1), ref['logits'], rtol=1e-3, atol=1e-5)
    g = model.get_grads()
    for n in O.dense_param_names(arch):
        rel_close(g[n], ref['grads'][n], 2e-3, n)
    rel_close(g['user_row_grads'], ref['user_row_grads'], 2e-3, 'user_row_grads')
    assert not g['user_row_grads'][~keep].any()
    if arch == 'iigru':
        rel_close(model.peek('duser2').cpu().numpy().reshape(B, -1), ref['user_row_grads2'], 2e-3, 'user_row_grads2')
    # the test model (inference) uses every user embedding unmasked
    X = [user.astype(np.int64), hist.astype(np.float64), cands[:, 0].astype(np.float64)]
    got = model.predict(X)[:, 0]
    want = O.predict(p, arch, user, hist, cands[:, 0])
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5)
