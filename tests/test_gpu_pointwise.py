"""GPU parity of the pointwise sibling (Seq2VecPaper / Seq2VecPaperDot / Seq2VecPaperId —
concat-MLP or dot scorer + the weighted BCE of task/seq2vec.py:213-216) against the CPU oracle, through the C-ABI.
Tolerances as in test_gpu_model.py."""
import numpy as np
import pytest

from conftest import make_batch
from test_gpu_model import rel_close, TINY, MID, ODD

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')


def build(arch, head, dims, B, W, T, K, drop, gain):
    from oracle import lstur_oracle as O
    from mnexp_amd.native_model import NativeModel
    rng = np.random.default_rng(11)
    p = O.init_params(rng, arch=arch, dtype=np.float64, head=head, **dims)
    for n in ('conv_b', 'att_b', 'dense_b', 'gru_b', 'score_b1', 'score_b2', 'uatt_b'):
        if n in p:
            p[n] = rng.normal(0, 0.1, size=p[n].shape)
    model = NativeModel(p['word_emb'], dims['N_u'], B, W, T, K, dims['F'], dims['k'], dims['U'], arch=arch,
                        dropout=drop, lr=1e-3, seed=77, head=head, gain=gain)
    model.set_weights({n: p[n] for n in p if n != 'word_emb'})
    return O, p, model, rng


@pytest.mark.parametrize('arch', ['igru', 'nigru', 'vo', 'niavg', 'niatt', 'ngru', 'iicgru'])
@pytest.mark.parametrize('head', ['point_mlp', 'point_dot'])
@pytest.mark.parametrize('dims,B,W,T,drop', [(TINY, 6, 4, 7, 0.0), (MID, 19, 9, 11, 0.2), (ODD, 7, 3, 6, 0.25)])
def test_pointwise_step_matches_oracle(arch, head, dims, B, W, T, drop):
    _need_gpu()
    if arch in ('ngru', 'iicgru') and head == 'point_dot':
        pytest.skip('a 2U-wide user vector has no dot scorer (paper.py:339-342 pairs it with the MLP)')
    gain, K = 1.7, 4
    O, p, model, rng = build(arch, head, dims, B, W, T, K, drop, gain)
    user, hist, cands = make_batch(rng, B, W, T, 1, dims['V'], dims['N_u'])
    cand = cands[:, 0]
    labels = (rng.random(B) < 0.3).astype(np.float32)
    labels[0] = 1.0
    user[1] = user[0]
    hist[2] = 0
    ref = O.point_forward_backward(p, arch, head, user, hist, cand, labels, gain, K, drop_p=drop, seed=77, step=3)
    from mnexp_amd._lib import lib, check
    u, h, c, y = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cand), model._dev_f32(labels)
    check(lib.lstur_point_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), y.data_ptr(), B, 3, 1.0,
                                           model.stats.data_ptr(), model._stream()), 'point fb')
    torch.cuda.synchronize()
    loss, acc = model.stats.tolist()
    np.testing.assert_allclose(model.peek('logits').cpu().numpy()[:B], ref['logits'], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(model.peek('probs').cpu().numpy()[:B], ref['probs'], rtol=1e-3, atol=1e-6)
    assert abs(loss - ref['loss']) <= 1e-4 * max(1.0, abs(ref['loss']))
    assert abs(acc - ref['acc']) < 1e-6
    g = model.get_grads()
    for n in O.dense_param_names(arch, head=head):
        rel_close(g[n].reshape(-1), ref['grads'][n].reshape(-1), 2e-3, n)
    if arch in ('igru', 'vo', 'ngru', 'iicgru'):
        rel_close(g['user_row_grads'], ref['user_row_grads'], 2e-3, 'user_row_grads')
    if arch == 'iicgru':      # Seq2VecPaperId's `iigru`: the second table's rows get the tail half of d(user vector)
        rel_close(model.peek('duser2').cpu().numpy()[:B * dims['U']], ref['user_row_grads2'].reshape(-1), 2e-3, 'user_row_grads2')
    # model.predict of the pointwise model = the scorer's sigmoid output without dropout
    pr = model.predict([user, hist, cand])
    np.testing.assert_allclose(pr.reshape(-1), O.point_predict(p, arch, head, user, hist, cand), rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize('arch', ['igru', 'iicgru'])
def test_pointwise_training_tracks_oracle_and_auc_metric(arch):
    """5 Adam steps: per-step loss and the updated parameters (for iicgru: both user tables) follow the oracle; the
    running tf.metrics.auc restatement equals a direct evaluation of its published definition on all predictions so far."""
    _need_gpu()
    head, dims, B, W, T, K, gain = 'point_mlp', MID, 24, 6, 9, 4, 1.0
    O, p, model, rng = build(arch, head, dims, B, W, T, K, 0.2, gain)
    pf = {n: a.astype(np.float32) for n, a in p.items()}
    tables = ['user_emb'] + (['user_emb2'] if arch == 'iicgru' else [])
    opt = O.Adam(pf, O.dense_param_names(arch, head=head) + tables, lr=1e-3)
    probs, labs = [], []
    for step in range(5):
        user, hist, cands = make_batch(rng, B, W, T, 1, dims['V'], dims['N_u'])
        labels = (rng.random(B) < 0.3).astype(np.float32)
        ref = O.point_train_step(pf, arch, head, opt, user, hist, cands[:, 0], labels, gain, K, drop_p=0.2, seed=77,
                                 step=step)
        model.train_on_batch([user, hist, cands[:, 0]], labels)
        torch.cuda.synchronize()
        assert abs(model.stats[0].item() - ref['loss']) <= 2e-3 * max(1.0, abs(ref['loss'])), step
        probs.append(ref['probs'])
        labs.append(labels)
    w = model.get_weights()
    for n in O.dense_param_names(arch, head=head) + tables:
        rel_close(w[n].reshape(-1), pf[n].reshape(-1), 2e-3, n)
    # tf.metrics.auc over everything seen (200 thresholds, trapezoidal), evaluated directly
    pr, y = np.concatenate(probs), np.concatenate(labs) > 0.5
    thr = np.array([-1e-7] + [(i + 1) / 199.0 for i in range(198)] + [1.0 + 1e-7])
    pred = pr[:, None] > thr[None, :]
    tp, fn = (pred & y[:, None]).sum(0), (~pred & y[:, None]).sum(0)
    fp, tn = (pred & ~y[:, None]).sum(0), (~pred & ~y[:, None]).sum(0)
    rec, fpr = (tp + 1e-6) / (tp + fn + 1e-6), fp / (fp + tn + 1e-6)
    want = float(np.sum((fpr[:-1] - fpr[1:]) * (rec[:-1] + rec[1:]) / 2))
    assert abs(model.auc.value() - want) < 5e-3       # GPU probabilities differ from the oracle's in the 4th digit


def test_head_argument_checks():
    _need_gpu()
    from mnexp_amd import _lib
    from mnexp_amd.native_model import NativeModel
    emb = np.zeros((20, 8), np.float32)
    with pytest.raises(_lib.LsturError, match='the pointwise heads support arch'):
        NativeModel(emb, 5, 4, 3, 5, 4, 8, 3, 6, arch='gru', head='point_mlp')
    m = NativeModel(emb, 5, 4, 3, 5, 4, 8, 3, 6, arch='igru', head='point_dot')
    z = torch.zeros(64, dtype=torch.int32, device='cuda')
    rc = _lib.lib.lstur_forward_backward(m._ctx, z.data_ptr(), z.data_ptr(), z.data_ptr(), 2, 0, 1.0,
                                        m.stats.data_ptr(), m._stream())
    assert rc == -3 and b'pointwise head' in _lib.lib.lstur_last_error()
