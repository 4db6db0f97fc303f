"""End-to-end training through the task/CLI boundary on the GPU, and AUC parity with the CPU oracle
(BASELINE.json: AUC within 0.002 of the CPU reference on the synthetic set).  Dropout is ON: the
counter-based masks are restated bit-exactly in the oracle, so both sides see the same masks."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')


def _config(tmp_path, **kw):
    from test_data_path import make_config
    from mnexp_amd import settings
    cfg = make_config(os.path.join(GOLD, 'tiny'), **kw)
    settings.public_config.update(output_model_path=str(tmp_path), log_dir=str(tmp_path))
    cfg.output_model_path = str(tmp_path)
    cfg.log_dir = str(tmp_path)
    return cfg


def per_impression_auc(task, score_fn, limit):
    from sklearn.metrics import roc_auc_score
    aucs = []
    for i, rows in zip(range(limit), task.test_gen_ids()):
        (user, hist, cands), labels = task._stack_ids(rows)
        aucs.append(roc_auc_score(labels, score_fn(user, hist, cands[:, 0])))
    return float(np.mean(aucs))


def test_training_tracks_oracle_and_auc_matches(tmp_path):
    _need_gpu()
    from mnexp_amd import task as T
    from oracle import lstur_oracle as O
    cfg = _config(tmp_path, dropout=0.2, batch_size=8, epochs=1, seed=5)
    h = T.get(cfg)
    model = h.build_model(0)
    tt = h.docs.title_table
    p = {n: a.astype(np.float32) for n, a in model.get_weights().items()}
    p['word_emb'] = model.word_emb.cpu().numpy()
    names = O.dense_param_names('igru') + ['user_emb']
    opt = O.Adam(p, names, lr=cfg.learning_rate)
    np.random.seed(3)
    random.seed(3)
    gen = h.train_ids
    steps = 60
    losses_gpu, losses_cpu = [], []
    for s in range(steps):
        (user, hist, cands), _ = next(gen)
        out = O.train_step(p, 'igru', opt, user, tt[hist], tt[cands], drop_p=0.2, seed=5, step=s)
        losses_cpu.append(out['loss'])
        model.stats_acc.zero_()
        model.train_on_ids(user, hist, cands)
        losses_gpu.append(model.stats_acc[0].item())
    np.testing.assert_allclose(losses_gpu, losses_cpu, rtol=2e-3, atol=2e-4)
    # same trend as the oracle's run (60 steps at lr 1e-3 move this tiny model's loss by ~1e-4 around ln 5, so the sign
    # of the change is not a robust property; the AUC agreement below and tools/e2e_train.py cover "it learns")
    trend = lambda l: np.mean(l[-10:]) - np.mean(l[:10])
    assert abs(trend(losses_gpu) - trend(losses_cpu)) < 3e-4
    auc_gpu = per_impression_auc(h, lambda u, hi, c: model.predict_ids(u, hi, c.reshape(-1, 1))[:, 0], 60)
    auc_cpu = per_impression_auc(h, lambda u, hi, c: O.predict(p, 'igru', u, tt[hi], tt[c]), 60)
    assert abs(auc_gpu - auc_cpu) <= 0.002, (auc_gpu, auc_cpu)


@pytest.mark.parametrize('arch,ids', [('igru', True), ('gru', False), ('hgru', True), ('igru', 'native')])
def test_cli_train_loop(tmp_path, arch, ids):
    _need_gpu()
    from mnexp_amd.main import run_training
    cfg = _config(tmp_path, arch=arch, epochs=2, training_step=15, validation_step=2, validation_impression=20,
                  testing_impression=20, dropout=0.2, ids=bool(ids), native_feed=(ids == 'native'), seed=1)
    np.random.seed(0)
    h = run_training(cfg)
    assert 0.0 <= h.last_metrics['auc'] <= 1.0 and np.isfinite(h.last_metrics['mrr'])
    assert abs(h.model.optimizer.lr - 1e-3 * 0.2 ** 2) < 1e-12              # lr decays every epoch
    assert h.model.optimizer.iterations == 30


@pytest.mark.parametrize('native_feed', [False, True])
def test_idless_task_with_the_native_feed(tmp_path, native_feed):
    """Seq2VecPaperSoftmax (no user column) through --ids and through --native-feed: the C++ feed yields (hist, cands)
    for an id-less task (it used to yield a user column the model then rejected)."""
    _need_gpu()
    from mnexp_amd.main import run_training
    cfg = _config(tmp_path, task='Seq2VecPaperSoftmax', arch='gru', epochs=1, training_step=12, validation_step=1,
                  validation_impression=10, testing_impression=10, dropout=0.2, seed=1, ids=True, native_feed=native_feed)
    np.random.seed(0)
    h = run_training(cfg)
    assert h.model.arch == 'nigru' and h.model.optimizer.iterations == 12 and 0.0 <= h.last_metrics['auc'] <= 1.0


@pytest.mark.parametrize('tname,arch', [('Seq2VecPaperSoftmaxDaysId', 'igru'), ('Seq2VecPaperSoftmaxDaysId', 'dgru'),
                                        ('Seq2VecPaperSoftmaxDays', 'gru')])
def test_cli_train_loop_days_variants(tmp_path, tname, arch):
    """The time-window variants (paper.py:668-881: history entries expire `days` after their click) driven through the
    native model: one epoch of the CLI loop incl. the evaluation callback.  `dgru` = the LSTUR paper's user-embedding
    masking (whole-row dropout 0.5, paper.py:608-611) needs a 2U-compatible scorer."""
    _need_gpu()
    from mnexp_amd.main import run_training
    kw = dict(score_model='ddot') if arch == 'dgru' else {}
    cfg = _config(tmp_path, task=tname, arch=arch, epochs=1, training_step=12, validation_step=2, validation_impression=15,
                  testing_impression=15, dropout=0.2, seed=1, ids=True, days=3, **kw)
    np.random.seed(0)
    h = run_training(cfg)
    assert h.model.optimizer.iterations == 12 and 0.0 <= h.last_metrics['auc'] <= 1.0 and np.isfinite(h.last_metrics['mrr'])
    # the window really expires entries: with days = 3 some history slots read as pad that a plain window keeps
    rows = [r for _, r in zip(range(50), h.train_gen_ids())]
    hist = np.stack([r[1] if h.has_user else r[0] for r in rows])
    assert (hist == 0).any()


def test_cli_train_loop_mhsa_encoder(tmp_path):
    """`--news-encoder mhsa --title-filter-shape F heads`: the build-defined multi-head self-attention encoder
    (BASELINE.json configs[4]; no reference counterpart) through the same task / CLI loop."""
    _need_gpu()
    from mnexp_amd.main import run_training
    cfg = _config(tmp_path, news_encoder='mhsa', title_filter_shape=(32, 2), epochs=2, training_step=15,
                  validation_step=2, validation_impression=20, testing_impression=20, dropout=0.2, ids=True, seed=1)
    np.random.seed(0)
    h = run_training(cfg)
    assert h.model.news_encoder == 'mhsa' and 'qkv_w' in h.model.param_names()
    assert 0.0 <= h.last_metrics['auc'] <= 1.0 and np.isfinite(h.last_metrics['mrr'])
    assert h.model.optimizer.iterations == 30
    with pytest.raises(Exception, match='not on the native path'):
        run_training(_config(tmp_path, news_encoder='gruatt', epochs=1))


@pytest.mark.parametrize('arch,score_model', [('dgru', 'dnn'), ('ngru', 'ddot'), ('igru', 'ddot')])
def test_cli_train_loop_score_models(tmp_path, arch, score_model):
    """`--score-model ddot | dnn` and the LSTUR-con user encoders that need them (`-a ngru | dgru`, paper.py:600-611)."""
    _need_gpu()
    from mnexp_amd.main import run_training
    cfg = _config(tmp_path, arch=arch, score_model=score_model, epochs=2, training_step=15, validation_step=2,
                  validation_impression=20, testing_impression=20, dropout=0.2, ids=True, seed=1)
    np.random.seed(0)
    h = run_training(cfg)
    assert h.model.score_model == score_model and h.model.arch == arch
    assert 0.0 <= h.last_metrics['auc'] <= 1.0 and np.isfinite(h.last_metrics['mrr'])
    assert h.model.optimizer.iterations == 30
    with pytest.raises(Exception, match='2U wide'):
        run_training(_config(tmp_path, arch='ngru', score_model='dot', epochs=1))


def test_cli_train_loop_trainable_word_table(tmp_path):
    """--textual-embedding-trainable (main.py:36, paper.py:132-138): the word table moves, its pad row included."""
    _need_gpu()
    from mnexp_amd.main import run_training
    cfg = _config(tmp_path, textual_embedding_trainable=True, epochs=1, training_step=10, validation_step=1,
                  validation_impression=10, testing_impression=10, dropout=0.2, ids=True, seed=1)
    np.random.seed(0)
    h = run_training(cfg)
    from mnexp_amd import utils
    emb0 = utils.load_textual_embedding(cfg.title_embedding_input, cfg.textual_embedding_dim)
    moved = np.abs(h.model.word_emb.cpu().numpy() - emb0)
    assert h.model.word_trainable and moved.max() > 1e-3 and moved[0].max() > 0


def test_idless_softmax_task(tmp_path):
    """Seq2VecPaperSoftmax (no user id, paper.py:384-524) with its `gru` user encoder."""
    _need_gpu()
    from mnexp_amd.main import run_training
    cfg = _config(tmp_path, task='Seq2VecPaperSoftmax', arch='gru', epochs=1, training_step=10, validation_step=1,
                  validation_impression=10, testing_impression=10, dropout=0.2, seed=1)
    np.random.seed(0)
    h = run_training(cfg)
    assert h.model.arch == 'nigru' and 0.0 <= h.last_metrics['auc'] <= 1.0
    for arch, native in (('att', 'niatt'), ('avg', 'niavg')):        # paper.py:207-215
        cfg = _config(tmp_path, task='Seq2VecPaperSoftmax', arch=arch, epochs=1, training_step=10, validation_step=1,
                      validation_impression=10, testing_impression=10, dropout=0.2, seed=1)
        np.random.seed(0)
        h = run_training(cfg)
        assert h.model.arch == native and 0.0 <= h.last_metrics['auc'] <= 1.0
    cfg = _config(tmp_path, task='Seq2VecPaperSoftmax', arch='igru', epochs=1)
    with pytest.raises(Exception, match='gru . avg . att'):
        run_training(cfg)


def test_click_entry_point(tmp_path):
    _need_gpu()
    from click.testing import CliRunner
    from mnexp_amd import settings
    from mnexp_amd import main as M  # noqa: F401  (registers the train command)
    np.random.seed(0)
    r = CliRunner().invoke(settings.main, [
        '--input-training-data-path', os.path.join(GOLD, 'tiny'), '--output-model-path', str(tmp_path),
        '--log-dir', str(tmp_path), 'train', '-t', 'Seq2VecPaperSoftmaxId', '-a', 'igru', '-e', '1', '-b', '8',
        '-s', '5', '-v', '1', '-cvi', '5', '-i', '5', '-w', '5', '--title-shape', '8', '--title-filter-shape', '16', '3',
        '--user-embedding-dim', '8', '--textual-embedding-dim', '16', '--name', 'x'], catch_exceptions=False)
    assert r.exit_code == 0, r.output
    assert os.path.exists(os.path.join(str(tmp_path), 'logx.txt'))


@pytest.mark.parametrize('arch', ['igru', 'iigru'])
def test_dp_call_sequence_on_one_rank_nccl(arch):
    """The data-parallel step (RCCL all-reduce of the flat bucket with the sparse user rows in its tail + Adam over the
    exchanged rows) with a 1-rank 'nccl' group must equal the plain single-GPU step; iigru: the tail carries the row
    gradients of BOTH user tables."""
    _need_gpu()
    import torch.distributed as dist
    from conftest import make_batch
    from test_gpu_model import build, MID
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        dims, B, W, T, K1 = MID, 16, 6, 9, 5
        outs = []
        for force in (False, True):
            O, p, model, rng = build(arch, dims, B, W, T, K1, 0.2)
            model.force_collectives = force
            for step in range(2):
                user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
                user[3] = user[0]
                X = [user.astype(np.int64), hist.astype(np.float64)] + [cands[:, j].astype(np.float64) for j in range(K1)]
                model.train_on_batch(X, None)
            torch.cuda.synchronize()
            outs.append(model.get_weights())
        for n in outs[0]:
            assert np.array_equal(outs[0][n], outs[1][n]), n
        assert ('user_emb2' in outs[0]) == (arch == 'iigru')
    finally:
        if created:
            dist.destroy_process_group()


def test_dp_tail_of_two_tables_concatenates_over_ranks():
    """lstur_dp_stage / lstur_dp_unpack for iigru: two 'ranks' (two contexts on this GPU) stage their rows and both tables'
    row gradients into their slot of the tail; the SUM of the two tails (what the all-reduce delivers) unpacks to the
    concatenation in rank order, and one Adam step over it equals the oracle's dense update of both tables."""
    _need_gpu()
    from conftest import make_batch
    from test_gpu_model import build, MID, rel_close
    from mnexp_amd._lib import lib, check
    import ctypes as C
    dims, B, W, T, K1 = MID, 12, 5, 9, 5
    world = 2
    models, tails, refs = [], [], []
    O = None
    for r in range(world):
        O, p, model, rng = build('iigru', dims, B, W, T, K1, 0.0, seed=5)      # identical replicas
        rb = np.random.default_rng(100 + r)
        user, hist, cands = make_batch(rb, B, W, T, K1, dims['V'], dims['N_u'])
        user[1] = user[0]
        u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
        check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 0, 1.0 / world,
                                         model.stats.data_ptr(), model._stream()))
        n_tail = int(lib.lstur_dp_tail_floats(C.byref(model.cfg), world))
        assert n_tail == world * B * (1 + 2 * dims['U'])
        tail = torch.zeros(n_tail, dtype=torch.float32, device='cuda')
        check(lib.lstur_dp_stage(model._ctx, tail.data_ptr(), r, world, model._stream()))
        models.append(model)
        tails.append(tail)
        refs.append(O.forward_backward(p, 'iigru', user, hist, cands))
        refs[-1]['user'] = user
    torch.cuda.synchronize()
    total = tails[0] + tails[1]                                                # the all-reduce
    m = models[0]
    rows, grads = C.c_void_p(), C.c_void_p()
    check(lib.lstur_dp_unpack(m._ctx, total.data_ptr(), world, B, C.byref(rows), C.byref(grads), m._stream()))
    got_rows = m._peek_ptr(rows.value, world * B, torch.int32).cpu().numpy()
    got_g1 = m._peek_ptr(grads.value, world * B * dims['U'], torch.float32).cpu().numpy().reshape(world * B, -1)
    assert np.array_equal(got_rows, np.concatenate([r['user'] for r in refs]))
    want_g1 = np.concatenate([r['user_row_grads'] for r in refs]) / world
    rel_close(got_g1, want_g1, 2e-3, 'table 1 rows')
    # one Adam step over the exchanged rows: both tables move like the oracle's dense update of the summed gradients
    before2 = m.user_emb2.clone()
    check(lib.lstur_adam_step(m._ctx, 1e-3, 1, rows.value, grads.value, world * B, m._stream()))
    torch.cuda.synchronize()
    want_g2 = np.concatenate([r['user_row_grads2'] for r in refs]) / world
    dense2 = O.densify_user_grad(dims['N_u'], dims['U'], got_rows, want_g2, np.float64)
    moved = (m.user_emb2 - before2).cpu().numpy().reshape(dims['N_u'], dims['U'])
    touched = np.abs(dense2).sum(1) > 0
    # first Adam step: delta = -lr_t * g / (|g| + eps') ~ -lr * sign(g) for the rows with a gradient, 0 elsewhere
    assert np.abs(moved[~touched]).max() == 0.0
    big = np.abs(dense2) > 1e-9
    assert np.array_equal(np.sign(moved[big]), -np.sign(dense2[big]))


def test_save_load_round_trip(tmp_path):
    _need_gpu()
    from conftest import make_batch
    from test_gpu_model import build, MID
    dims, B, W, T, K1 = MID, 8, 5, 9, 5
    O, p, model, rng = build('gru', dims, B, W, T, K1, 0.0)
    paths = (str(tmp_path / 'm.json'), str(tmp_path / 'm.pkl'))
    model.save(paths)
    O2, p2, other, _ = build('gru', dims, B, W, T, K1, 0.0, seed=99)     # different weights
    user, hist, cands = make_batch(rng, 6, W, T, 1, dims['V'], dims['N_u'])
    X = [user.astype(np.int64), hist.astype(np.float64), cands[:, 0].astype(np.float64)]
    a = model.predict(X)
    assert not np.allclose(a, other.predict(X))
    other.load(paths)
    assert np.array_equal(a, other.predict(X))
    import pickle
    arrays = pickle.load(open(paths[1], 'rb'))
    assert arrays[2].shape == (dims['F'], 1) and arrays[-1].shape == (dims['V'], dims['D'])   # Keras shapes / order


@pytest.mark.parametrize('tname,arch,ids', [('Seq2VecPaperId', 'igru', False), ('Seq2VecPaperId', 'igru', True),
                                            ('Seq2VecPaperId', 'iigru', True), ('Seq2VecPaperId', 'gru', False),
                                            ('Seq2VecPaper', 'gru', False), ('Seq2VecPaperDot', 'gru', True)])
def test_pointwise_cli_train_loop(tmp_path, tname, arch, ids):
    """the pointwise sibling through the reference's train loop (fit_generator / callback of
    task/seq2vec.py:296-321 / evaluate_generator with the auc_roc metric)."""
    _need_gpu()
    from mnexp_amd.main import run_training
    cfg = _config(tmp_path, task=tname, arch=arch, epochs=2, training_step=15, validation_step=2,
                  validation_impression=20, testing_impression=20, dropout=0.2, ids=ids, seed=1, max_impression=3,
                  gain=1.5)
    np.random.seed(0)
    random.seed(0)
    h = run_training(cfg)
    assert h.model.pointwise and h.model.metrics_names == ['loss', 'auc_roc']
    assert h.model.head == ('point_dot' if tname.endswith('Dot') else 'point_mlp')
    assert 0.0 <= h.last_metrics['auc'] <= 1.0 and np.isfinite(h.last_metrics['mrr'])
    assert abs(h.model.optimizer.lr - 1e-3 * 0.2 ** 2) < 1e-12
    assert h.model.optimizer.iterations == 30
    assert abs(h.model.cfg.gain - 1.5) < 1e-7 and h.model.cfg.loss_negatives == cfg.negative_samples
    if tname == 'Seq2VecPaperId' and arch != 'igru':      # paper.py:339-348: 2U-wide concatenations, `iigru` with a second table
        assert h.model.arch == {'gru': 'ngru', 'iigru': 'iicgru'}[arch]
        assert h.model.shapes['score_w1'][0] == 3 * cfg.user_embedding_dim


def test_pointwise_training_through_the_task_tracks_oracle(tmp_path):
    """40 steps of Seq2VecPaperId batches (pool-shuffled, 1 positive : K negatives): per-step loss follows the
    oracle and the mean per-impression AUC of the trained scorers agrees to 0.002."""
    _need_gpu()
    from mnexp_amd import task as T
    from oracle import lstur_oracle as O
    cfg = _config(tmp_path, task='Seq2VecPaperId', arch='igru', dropout=0.2, batch_size=8, epochs=1, seed=5,
                  max_impression=3)
    h = T.get(cfg)
    model = h.build_model(0)
    tt = h.docs.title_table
    p = {n: a.astype(np.float32) for n, a in model.get_weights().items()}
    p['word_emb'] = model.word_emb.cpu().numpy()
    names = O.dense_param_names('igru', head='point_mlp') + ['user_emb']
    opt = O.Adam(p, names, lr=cfg.learning_rate)
    np.random.seed(3)
    random.seed(3)
    gen = h.train_ids
    K = cfg.negative_samples
    losses_gpu, losses_cpu = [], []
    for s in range(40):
        (user, hist, cands), labels = next(gen)
        out = O.point_train_step(p, 'igru', 'point_mlp', opt, user, tt[hist], tt[cands[:, 0]], labels, 1.0, K,
                                 drop_p=0.2, seed=5, step=s)
        losses_cpu.append(out['loss'])
        model.stats_acc.zero_()
        model.train_on_ids(user, hist, cands, labels=labels)
        losses_gpu.append(model.stats_acc[0].item())
    np.testing.assert_allclose(losses_gpu, losses_cpu, rtol=2e-3, atol=2e-4)
    auc_gpu = per_impression_auc(h, lambda u, hi, c: model.predict_ids(u, hi, c.reshape(-1, 1))[:, 0], 60)
    auc_cpu = per_impression_auc(h, lambda u, hi, c: O.point_predict(p, 'igru', 'point_mlp', u, tt[hi], tt[c]), 60)
    assert abs(auc_gpu - auc_cpu) <= 0.002, (auc_gpu, auc_cpu)


def test_single_collective_dp_with_two_emulated_ranks():
    """The data-parallel step with world = 2 without a second GPU: two models play rank 0 / rank 1 (same weights,
    different batches), each stages its sparse user rows into its slot of the bucket tail, the buckets are summed
    (what the ONE all-reduce does), both unpack and apply Adam.  Result = the oracle's step on the concatenated
    global batch, and the two replicas stay bit-identical."""
    _need_gpu()
    import ctypes as C
    from conftest import make_batch
    from mnexp_amd._lib import lib, check
    from test_gpu_model import build, MID, rel_close
    dims, B, W, T, K1 = MID, 12, 6, 9, 5
    O, p, m0, rng = build('igru', dims, B, W, T, K1, 0.0)
    _, _, m1, _ = build('igru', dims, B, W, T, K1, 0.0)
    batches = [make_batch(rng, B, W, T, K1, dims['V'], dims['N_u']) for _ in range(2)]
    batches[1][0][:3] = batches[0][0][:3]                     # the same users on both ranks: duplicates must sum
    world, dense = 2, m0.sizes.dense
    used = int(lib.lstur_dp_tail_floats(C.byref(m0.cfg), world))
    for r, m in enumerate((m0, m1)):
        user, hist, cands = batches[r]
        u, h, c = m._dev_i32(user), m._dev_i32(hist), m._dev_i32(cands)
        check(lib.lstur_forward_backward(m._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 0, 1.0 / world,
                                         m.stats.data_ptr(), m._stream()), 'fb')
        check(lib.lstur_dp_stage(m._ctx, m._bucket[dense:].data_ptr(), r, world, m._stream()), 'stage')
    total = m0._bucket[:dense + used] + m1._bucket[:dense + used]       # the all-reduce
    for m in (m0, m1):
        m._bucket[:dense + used].copy_(total)
        rows, grads = C.c_void_p(), C.c_void_p()
        check(lib.lstur_dp_unpack(m._ctx, m._bucket[dense:].data_ptr(), world, B, C.byref(rows), C.byref(grads),
                                  m._stream()), 'unpack')
        check(lib.lstur_adam_step(m._ctx, 1e-3, 1, rows.value, grads.value, world * B, m._stream()), 'adam')
    torch.cuda.synchronize()
    w0, w1 = m0.get_weights(), m1.get_weights()
    for n in w0:
        assert np.array_equal(w0[n], w1[n]), n                 # replicas bit-identical
    pf = {n: a.astype(np.float32) for n, a in p.items()}
    opt = O.Adam(pf, O.dense_param_names('igru') + ['user_emb'], lr=1e-3)
    user = np.concatenate([b[0] for b in batches])
    hist = np.concatenate([b[1] for b in batches])
    cands = np.concatenate([b[2] for b in batches])
    O.train_step(pf, 'igru', opt, user, hist, cands, drop_p=0.0)
    for n in O.dense_param_names('igru') + ['user_emb']:
        rel_close(w0[n].reshape(-1), pf[n].reshape(-1), 2e-3, n)


def test_pretrained_frozen_news_encoder(tmp_path):
    """--enable-pretrain-encoder (paper.py:103-107): the news encoder is loaded from the encoder files and, without
    --pretrain-encoder-trainable, stays fixed while the user encoder trains; gradients equal the oracle's for the
    parameters that do train."""
    _need_gpu()
    from conftest import make_batch
    from mnexp_amd.native_model import NativeModel
    from test_gpu_model import build, MID, rel_close
    dims, B, W, T, K1 = MID, 10, 6, 9, 5
    O, p, donor, rng = build('igru', dims, B, W, T, K1, 0.2)
    paths = (str(tmp_path / 'encoder.json'), str(tmp_path / 'encoder.pkl'))
    donor.save_encoder(paths)
    import pickle
    arrays = pickle.load(open(paths[1], 'rb'))
    assert [a.shape for a in arrays][:4] == [(dims['V'], dims['D']), (dims['k'], dims['D'], dims['F']), (dims['F'],),
                                              (dims['F'], 1)]                     # doc_encoder.get_weights() order
    rng2 = np.random.default_rng(99)
    model = NativeModel(rng2.normal(0, 0.1, (dims['V'], dims['D'])), dims['N_u'], B, W, T, K1 - 1, dims['F'], dims['k'],
                        dims['U'], arch='igru', dropout=0.2, lr=1e-2, seed=77, freeze_encoder=True)
    model.load_encoder(paths)
    enc = ['conv_w', 'conv_b', 'att_w', 'att_b', 'dense_w', 'dense_b']
    model.set_weights({n: p[n] for n in ('user_emb', 'gru_w', 'gru_u', 'gru_b')})
    w_before = model.get_weights()
    for n in enc:
        assert np.array_equal(w_before[n], p[n].astype(np.float32)), n
    assert np.array_equal(model.word_emb.cpu().numpy(), p['word_emb'].astype(np.float32))
    user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
    ref = O.forward_backward(p, 'igru', user, hist, cands, drop_p=0.2, seed=77, step=0)
    X = [user.astype(np.int64), hist.astype(np.float64)] + [cands[:, j].astype(np.float64) for j in range(K1)]
    model.train_on_batch(X, None)
    torch.cuda.synchronize()
    assert abs(model.stats[0].item() - ref['loss']) < 1e-4 * max(1.0, abs(ref['loss']))
    g = model.get_grads()
    for n in ('gru_w', 'gru_u', 'gru_b'):
        rel_close(g[n].reshape(-1), ref['grads'][n].reshape(-1), 2e-3, n)
    for n in enc:
        assert not g[n].any(), n                                # no gradient behind the doc vectors
    for _ in range(3):
        model.train_on_batch(X, None)
    torch.cuda.synchronize()
    w_after = model.get_weights()
    for n in enc:
        assert np.array_equal(w_after[n], w_before[n]), n       # frozen
    assert not np.array_equal(w_after['gru_w'], w_before['gru_w'])   # the user encoder learns


def test_cli_with_pretrained_encoder(tmp_path):
    _need_gpu()
    from mnexp_amd.main import run_training
    from mnexp_amd import settings
    cfg = _config(tmp_path, arch='igru', epochs=1, training_step=8, validation_step=1, validation_impression=10,
                  testing_impression=10, dropout=0.2, seed=1)
    h = run_training(cfg)
    settings.public_config.update(input_previous_model_path=str(tmp_path))
    cfg2 = _config(tmp_path, arch='igru', epochs=1, training_step=8, validation_step=1, validation_impression=10,
                   testing_impression=10, dropout=0.2, seed=2, enable_pretrain_encoder=True, pretrain_name='')
    cfg2.input_previous_model_path = str(tmp_path)
    h.model.save_encoder(cfg2.encoder_input)
    h2 = run_training(cfg2)
    assert h2.model.freeze_encoder
    a, b = h.model.get_weights(), h2.model.get_weights()
    for n in ('conv_w', 'att_w', 'dense_w', 'dense_b'):
        assert np.array_equal(a[n], b[n]), n
    assert not np.array_equal(a['gru_u'], b['gru_u'])
