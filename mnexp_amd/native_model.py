"""Keras-like model object backed by liblstur_hip.so (HIP kernels for gfx950).

Stands in for the ``keras.Model`` that ``Seq2VecPaperSoftmaxId._build_model`` compiles
(/root/reference/task/paper.py:635-665) and exposes exactly what the reference's callers use:
``fit_generator`` / ``evaluate_generator`` (main.py:73-87), ``predict`` (task/seq2vec.py:206),
``metrics_names``, ``optimizer.lr`` (task/paper.py:498-499).  PyTorch-ROCm supplies device
memory, the stream and (under data parallelism) RCCL collectives; all arithmetic of the model
runs in the library.  There is no CPU path.
"""
import ctypes as C
import os
import logging
import pickle

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def prefetched(generator, count, depth=10):
    """The next ``count`` items of ``generator``, drawn by one background thread at most ``depth`` ahead (what Keras'
    GeneratorEnqueuer does for fit_generator with workers=1); an exception in the generator is re-raised here."""
    import queue
    import threading
    q = queue.Queue(max(1, int(depth)))

    def work():
        try:
            for _ in range(count):
                q.put((True, next(generator)))
        except BaseException as e:          # noqa: B902 -- handed to the consumer
            q.put((False, e))

    t = threading.Thread(target=work, daemon=True)
    t.start()
    for _ in range(count):
        ok, item = q.get()
        if not ok:
            raise item
        yield item
    t.join()


class History:
    def __init__(self):
        self.history = {}


class _WeightsUnpickler(pickle.Unpickler):
    """Weights-only reader of the ``pickle(model.get_weights())`` files of utils.save_model (utils.py:74-79): the
    payload is a list of ndarrays, so only numpy's array reconstruction globals are admitted.  Any other global --
    what a crafted or foreign pickle would need to run code -- is refused."""
    _ALLOWED = {('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
                ('numpy', 'ndarray'), ('numpy', 'dtype'),
                ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
                ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer'),
                ('_codecs', 'encode')}       # protocol-2 pickles carry the array bytes as a latin-1 str

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError('weights file references {}.{}: only numpy arrays are accepted'.format(module, name))


def load_weight_list(path):
    """The list of ndarrays in a weights pickle; executes nothing from the file."""
    with open(path, 'rb') as f:
        arrays = _WeightsUnpickler(f).load()
    if not isinstance(arrays, (list, tuple)) or not all(isinstance(a, np.ndarray) and a.dtype != object for a in arrays):
        raise ValueError('{}: expected a list of numeric ndarrays'.format(path))
    return list(arrays)


class _Optimizer:
    """Carrier of the learning rate (``model.optimizer.lr`` get/set, task/paper.py:498-499)."""

    def __init__(self, lr):
        self.lr = float(lr)
        self.iterations = 0


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _orthogonal(rng, rows, cols):
    a = rng.normal(0.0, 1.0, size=(rows, cols))
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == (rows, cols) else vt
    return np.ascontiguousarray(q[:rows, :cols], dtype=np.float32)


class _TfAuc:
    """``utils.auc_roc`` of the pointwise models (utils.py:88-103) = ``tf.metrics.auc(y_true, y_pred)`` with its
    defaults: 200 thresholds, ROC curve, trapezoidal summation, confusion counts ACCUMULATED over the calls until the
    metric variables are re-initialised (task/seq2vec.py:296-298 does that once per epoch).  Restated from the
    published TF-1 definition; counts live on the device so a training step never synchronises."""

    def __init__(self, device, num_thresholds=200):
        k = 1e-7
        thr = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
        self.thr = torch.tensor([0.0 - k] + thr + [1.0 + k], dtype=torch.float32, device=device)
        self.counts = torch.zeros(4, num_thresholds, dtype=torch.float64, device=device)   # tp, fn, tn, fp

    def reset(self):
        self.counts.zero_()

    def update(self, labels, probs):
        y = labels.reshape(-1, 1) > 0.5
        pred = probs.reshape(-1, 1) > self.thr.reshape(1, -1)
        self.counts[0] += (pred & y).sum(0)
        self.counts[1] += (~pred & y).sum(0)
        self.counts[2] += (~pred & ~y).sum(0)
        self.counts[3] += (pred & ~y).sum(0)

    def value(self):
        tp, fn, tn, fp = self.counts.cpu().numpy()
        eps = 1e-6
        rec = (tp + eps) / (tp + fn + eps)
        fpr = fp / (fp + tn + eps)
        return float(np.sum((fpr[:-1] - fpr[1:]) * (rec[:-1] + rec[1:]) / 2.0))


class NativeModel:
    def __init__(self, word_emb, n_users, batch_size, window, title_len, negative_samples, filters, kernel,
                 user_dim, arch='igru', dropout=0.2, lr=1e-3, seed=0, device=None, title_table=None,
                 dist_group=None, has_user=True, head='softmax', gain=1.0, freeze_encoder=False,
                 word_trainable=False, news_encoder='cnnatt', heads=0, score_model='dot', id_keep=1.0):
        if not torch.cuda.is_available():
            raise _lib.LsturError('no GPU visible: the native LSTUR model has no CPU fallback')
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        word_emb = np.ascontiguousarray(word_emb, dtype=np.float32)
        V, D = word_emb.shape
        self.head = head
        self.pointwise = head != 'softmax'          # one candidate + one label per sample
        self.W, self.T, self.B = window, title_len, batch_size
        self.K1 = 1 if self.pointwise else negative_samples + 1
        self.metrics_names = ['loss', 'auc_roc'] if self.pointwise else ['loss', 'categorical_accuracy']
        self.arch = arch
        self.has_user = has_user       # False: id-less task (Seq2VecPaperSoftmax), inputs carry no user column
        # dropout streams are keyed by (seed, step, stream): fold the data-parallel rank into the seed, or every rank would
        # drop the same element positions of its own samples (weights are still initialised from `seed` on every rank)
        rank = torch.distributed.get_rank(dist_group) if (torch.distributed.is_available() and
                                                          torch.distributed.is_initialized()) else 0
        self.cfg = _lib.make_cfg(batch_size, window, title_len, self.K1, V, D, filters, kernel, user_dim,
                                 max(1, n_users), arch, dropout, (int(seed) + 0x9E3779B1 * rank) & 0xFFFFFFFF, head=head, gain=gain,
                                 loss_negatives=max(1, negative_samples), freeze_encoder=freeze_encoder,
                                 word_trainable=word_trainable, news_encoder=news_encoder, heads=heads,
                                 score_model=score_model, id_keep=id_keep)
        self.id_keep = float(id_keep)        # --id-keep: user-embedding masking probability in training (cook.py:139-142)
        self.score_model = score_model       # listwise --score-model dot | ddot | dnn (paper.py:443-458)
        self.news_encoder = news_encoder     # 'mhsa': build-defined multi-head self-attention token stage (heads x F/heads)
        self.mhsa_proj_dtype = 'bf16'        # operand type of the mhsa Q|K|V projection GEMMs (attention products: bf16)
        self.freeze_encoder = bool(freeze_encoder)
        self.word_trainable = bool(word_trainable)     # --textual-embedding-trainable (paper.py:132-138)
        self.sizes = _lib.query_sizes(self.cfg)
        self.layout = _lib.param_layout(self.cfg)
        H, Ue = self.sizes.gru_dim, self.sizes.user_emb_dim
        self.shapes = dict(conv_w=(kernel, D, filters), conv_b=(filters,), qkv_w=(D, 3 * filters), qkv_b=(3 * filters,),
                           att_w=(filters,), att_b=(1,),
                           dense_w=(filters, user_dim), dense_b=(user_dim,), gru_w=(user_dim, 3 * H),
                           gru_u=(H, 3 * H), gru_b=(3 * H,), con_w=(H + Ue, user_dim), con_b=(user_dim,),
                           user_emb=(max(1, n_users), Ue), word_emb=(V, D),
                           score_b1=(user_dim,), score_w2=(user_dim,),
                           score_b2=(1,))
        Uu = 2 * user_dim if arch in ('ngru', 'dgru', 'iicgru') else user_dim      # width of the user vector
        self.shapes.update(score_w1=(Uu + user_dim, user_dim), sdu_w=(Uu, user_dim), sdu_b=(user_dim,), sdd_w=(user_dim, user_dim), sdd_b=(user_dim,),
                           sn1_w=(Uu + user_dim, user_dim), sn1_b=(user_dim,), sn2_w=(user_dim,), sn2_b=(1,),
                           uatt_w=(user_dim,), uatt_b=(1,))
        dev = self.device
        f32 = torch.float32
        self.word_emb = torch.from_numpy(word_emb).to(dev)
        self.dense_param = torch.zeros(self.sizes.dense, dtype=f32, device=dev)
        # gradient bucket + room for the data-parallel exchange of the sparse user rows (up to 8 ranks): ONE all-reduce
        # of this buffer per step averages the dense gradients and concatenates the rows (include/lstur.h: lstur_dp_*)
        self._dp_tail = int(lib.lstur_dp_tail_floats(C.byref(self.cfg), 8))
        self._bucket = torch.zeros(self.sizes.dense + self._dp_tail, dtype=f32, device=dev)
        self.dense_grad = self._bucket[:self.sizes.dense]
        self.dense_m = torch.zeros(self.sizes.dense, dtype=f32, device=dev)
        self.dense_v = torch.zeros(self.sizes.dense, dtype=f32, device=dev)
        self.user_emb = torch.zeros(self.sizes.user_emb, dtype=f32, device=dev)
        self.user_m = torch.zeros(self.sizes.user_emb, dtype=f32, device=dev)
        self.user_v = torch.zeros(self.sizes.user_emb, dtype=f32, device=dev)
        self.workspace = torch.zeros(self.sizes.workspace_bytes + 256, dtype=torch.uint8, device=dev)
        ws_ptr = (self.workspace.data_ptr() + 255) // 256 * 256
        self._stats4 = torch.zeros(4, dtype=f32, device=dev)   # [loss, metric] of the last batch | running sums
        self.stats = self._stats4[:2]
        self.stats_acc = self._stats4[2:]
        wg = wm = wv = None
        if self.word_trainable:
            self.word_grad = torch.zeros(V * D, dtype=f32, device=dev)
            self.word_m = torch.zeros(V * D, dtype=f32, device=dev)
            self.word_v = torch.zeros(V * D, dtype=f32, device=dev)
            wg, wm, wv = self.word_grad.data_ptr(), self.word_m.data_ptr(), self.word_v.data_ptr()
        u2 = m2 = v2 = None
        if arch in ('iigru', 'iicgru'):   # the second user table of paper.py:617-618 and its Adam moments
            self.user_emb2 = torch.zeros(self.sizes.user_emb, dtype=f32, device=dev)
            self.user_m2 = torch.zeros(self.sizes.user_emb, dtype=f32, device=dev)
            self.user_v2 = torch.zeros(self.sizes.user_emb, dtype=f32, device=dev)
            u2, m2, v2 = self.user_emb2.data_ptr(), self.user_m2.data_ptr(), self.user_v2.data_ptr()
            self.shapes['user_emb2'] = self.shapes['user_emb']
        bufs = _lib.Buffers(self.word_emb.data_ptr(), self.dense_param.data_ptr(), self.dense_grad.data_ptr(),
                            self.dense_m.data_ptr(), self.dense_v.data_ptr(), self.user_emb.data_ptr(),
                            self.user_m.data_ptr(), self.user_v.data_ptr(), ws_ptr, self.sizes.workspace_bytes,
                            wg, wm, wv, u2, m2, v2)
        self._ctx = C.c_void_p()
        check(lib.lstur_create(C.byref(self.cfg), C.byref(bufs), C.byref(self._ctx)), 'lstur_create')
        # how the Conv1D contractions run, as the LIBRARY decides it: 'bf16x3' = fp32-grade 3-way split on the bf16 matrix
        # cores (convx.hip), 'f16x2' = the same with two fp16 parts / three products (the default), 'f32' = exact-fp32 MFMA
        # (conv.hip; filters > 448 or LSTUR_CONV_F32=1), 'n/a' = mhsa encoder
        self.conv_scheme = {0: 'n/a', 1: 'bf16x3', 2: 'f32', 3: 'f16x2'}[lib.lstur_conv_scheme(self._ctx)]
        self.lazy_table = bool(lib.lstur_lazy_adam(self._ctx))   # large user tables: dense Adam applied per row when it is next read
        self.optimizer = _Optimizer(lr)
        self.step = 0                       # dropout stream / fit step counter
        self.dist_group = dist_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(dist_group)
        # test hook: run the RCCL collectives even with one rank (exercises the exact DP call sequence)
        self.force_collectives = False
        self._dp_attached = None            # decided by the first data-parallel step (_dp_prepare)
        self.title_table = None
        if title_table is not None:
            self.set_title_table(title_table)
        self.auc = _TfAuc(dev) if self.pointwise else None
        self._init_weights(np.random.RandomState(seed))

    # ---- construction from a task handler ----------------------------------------
    @classmethod
    def from_task(cls, task, user_count):
        cfg = task.config
        from . import utils
        enc = getattr(cfg, 'news_encoder', 'cnnatt')
        if enc not in ('cnnatt', 'mhsa'):
            # the reference's gruatt / lstmatt / bigruatt / bilstmatt (paper.py:119-130) are out of scope; `mhsa` is the
            # build-defined multi-head self-attention encoder of BASELINE.json configs[4] (--title-filter-shape F heads)
            raise _lib.LsturError('news encoder {!r} is not on the native path (have cnnatt, mhsa)'.format(enc))
        head = getattr(task, 'head', 'softmax')
        score_model = getattr(cfg, 'score_model', 'dot') if head == 'softmax' else 'dot'
        emb = utils.load_textual_embedding(cfg.title_embedding_input, cfg.textual_embedding_dim,
                                           native=bool(getattr(cfg, 'native_load', False)))
        vmax = int(task.docs.title_table.max())
        if vmax >= emb.shape[0]:
            raise IndexError('title token id {} outside the {}-row word table'.format(vmax, emb.shape[0]))
        filters, kernel = cfg.title_filter_shape
        arch, has_user = cfg.arch, user_count > 0
        if not has_user:
            # Seq2VecPaper(Softmax).get_user_encoder (paper.py:198-221): only its `gru` (a plain masked GRU from a zero
            # state) is on the native path; it is the graph SoftmaxId calls `nigru`
            idless = {'gru': 'nigru', 'avg': 'niavg', 'att': 'niatt'}    # paper.py:206-215
            if arch not in idless:
                raise _lib.LsturError('id-less task: -a gru | avg | att (got {!r})'.format(arch))
            arch = idless[arch]
        elif head != 'softmax':
            # Seq2VecPaperId.get_user_encoder (paper.py:330-354): its `gru` is the plain concat [GRU(x) | user_vec] (2U wide,
            # the library's ngru) in front of the MLP scorer; its `iigru` is that concat behind an igru with a row of a SECOND
            # table and no Dense (paper.py:343-348: the library's iicgru)
            if arch == 'gru' and head == 'point_mlp':
                arch = 'ngru'
            elif arch == 'iigru' and head == 'point_mlp':
                arch = 'iicgru'
            elif arch not in ('igru', 'vo'):
                raise _lib.LsturError('pointwise task with user ids: -a igru | vo | gru | iigru (the last two with the MLP '
                                      'scorer) are on the native path (got {!r})'.format(arch))
        pretrained = bool(getattr(cfg, 'enable_pretrain_encoder', False))
        model = cls(emb, user_count, cfg.batch_size, cfg.window_size, cfg.title_shape, cfg.negative_samples,
                    int(filters), int(kernel), cfg.user_embedding_dim, arch=arch, dropout=cfg.dropout,
                    lr=cfg.learning_rate, seed=int(getattr(cfg, 'seed', 0) or 0),
                    title_table=task.docs.title_table, has_user=has_user, head=head,
                    gain=float(getattr(cfg, 'gain', 1.0)),
                    freeze_encoder=pretrained and not getattr(cfg, 'pretrain_encoder_trainable', False),
                    word_trainable=bool(getattr(cfg, 'textual_embedding_trainable', False)) and not pretrained,
                    news_encoder=enc, heads=int(kernel) if enc == 'mhsa' else 0, score_model=score_model,
                    id_keep=float(getattr(cfg, 'id_keep', 1.0) or 1.0))
        if pretrained:                      # paper.py:103-107
            model.load_encoder(cfg.encoder_input)
        return model

    def test_view(self):
        """The reference swaps ``model``/``test_model`` around evaluation (task/paper.py:502,524); here one
        object serves both roles because ``predict`` already has the test model's semantics."""
        return self

    # ---- weights -------------------------------------------------------------------
    def _init_weights(self, rng):
        """Keras initialisers of the layers the reference instantiates: glorot_uniform kernels, zero
        biases, orthogonal recurrent kernel, uniform(+-0.05) embedding."""
        s = self.shapes
        k, D, F = s['conv_w']
        U = s['dense_w'][1]
        H = s['gru_u'][0]
        w = dict(att_w=_glorot(rng, s['att_w'], F, 1), att_b=np.zeros(1, np.float32),
                 dense_w=_glorot(rng, s['dense_w'], F, U), dense_b=np.zeros(U, np.float32),
                 gru_w=_glorot(rng, s['gru_w'], U, 3 * H), gru_u=_orthogonal(rng, H, 3 * H),
                 gru_b=np.zeros(3 * H, np.float32),
                 user_emb=rng.uniform(-0.05, 0.05, size=s['user_emb']).astype(np.float32))
        if self.news_encoder == 'mhsa':    # three glorot_uniform [D, F] projections side by side, zero bias
            w['qkv_w'] = np.concatenate([_glorot(rng, (D, F), D, F) for _ in range(3)], axis=1)
            w['qkv_b'] = np.zeros(3 * F, np.float32)
        else:
            w['conv_w'] = _glorot(rng, s['conv_w'], k * D, k * F)
            w['conv_b'] = np.zeros(F, np.float32)
        if self.arch in ('gru', 'iigru'):
            w['con_w'] = _glorot(rng, s['con_w'], s['con_w'][0], U)
            w['con_b'] = np.zeros(U, np.float32)
        if self.arch in ('iigru', 'iicgru'):
            w['user_emb2'] = rng.uniform(-0.05, 0.05, size=s['user_emb2']).astype(np.float32)
        if self.arch == 'niatt':         # SimpleAttentionMaskSupport over the history (paper.py:207-209, models.py:448-449)
            w['uatt_w'] = _glorot(rng, s['uatt_w'], U, 1)
            w['uatt_b'] = np.zeros(1, np.float32)
        if self.score_model == 'ddot':   # Dense(U, tanh) on both sides (paper.py:452-454)
            w['sdu_w'] = _glorot(rng, s['sdu_w'], s['sdu_w'][0], U)
            w['sdu_b'] = np.zeros(U, np.float32)
            w['sdd_w'] = _glorot(rng, s['sdd_w'], U, U)
            w['sdd_b'] = np.zeros(U, np.float32)
        elif self.score_model == 'dnn':  # Dense(U, relu) -> Dense(1) on the concatenation (paper.py:448-451)
            w['sn1_w'] = _glorot(rng, s['sn1_w'], s['sn1_w'][0], U)
            w['sn1_b'] = np.zeros(U, np.float32)
            w['sn2_w'] = _glorot(rng, s['sn2_w'], U, 1)
            w['sn2_b'] = np.zeros(1, np.float32)
        if self.head == 'point_mlp':     # concat_dense / socre_dense (paper.py:223-227)
            w['score_w1'] = _glorot(rng, s['score_w1'], s['score_w1'][0], U)
            w['score_b1'] = np.zeros(U, np.float32)
            w['score_w2'] = _glorot(rng, s['score_w2'], U, 1)
            w['score_b2'] = np.zeros(1, np.float32)
        self.set_weights(w)

    def param_names(self):
        return [n for n, _, _ in self.layout] + ['user_emb'] + (['user_emb2'] if self.arch in ('iigru', 'iicgru') else [])

    def flush_tables(self):
        """Large user tables are updated lazily (include/lstur.h: lstur_flush_tables): bring every row up to date before
        the host reads or overwrites user_emb / user_m / user_v."""
        if getattr(self, '_ctx', None):
            check(lib.lstur_flush_tables(self._ctx, self._stream()), 'lstur_flush_tables')

    def set_weights(self, weights):
        """weights: dict name -> ndarray (any subset of param_names() + 'word_emb')."""
        self.flush_tables()
        for name, arr in weights.items():
            a = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)).to(self.device)
            if name == 'user_emb':
                self.user_emb.copy_(a)
            elif name == 'user_emb2':
                self.user_emb2.copy_(a)
            elif name == 'word_emb':
                self.word_emb.view(-1).copy_(a)
            else:
                off, cnt = self._slot(name)
                if a.numel() != cnt:
                    raise ValueError('{}: expected {} values, got {}'.format(name, cnt, a.numel()))
                self.dense_param[off:off + cnt].copy_(a)
        if getattr(self, '_ctx', None):      # the library's weight images / a staged batch are stale now
            check(lib.lstur_params_changed(self._ctx), 'lstur_params_changed')

    def _slot(self, name):
        for n, off, cnt in self.layout:
            if n == name:
                return off, cnt
        raise KeyError(name)

    def _get(self, flat, name):
        if name in ('user_emb', 'user_emb2'):
            self.flush_tables()            # (a lazily updated table: every row current before the host reads it)
        if name == 'user_emb':
            return self.user_emb.cpu().numpy().reshape(self.shapes['user_emb'])
        if name == 'user_emb2':
            return self.user_emb2.cpu().numpy().reshape(self.shapes['user_emb2'])
        off, cnt = self._slot(name)
        return flat[off:off + cnt].cpu().numpy().reshape(self.shapes[name])

    def get_weights(self):
        self.flush_tables()
        return {n: self._get(self.dense_param, n) for n in self.param_names()}

    def get_grads(self):
        """Dense gradients of the last train step by name (+ 'user_rows', 'user_row_grads')."""
        g = {n: self._get(self.dense_grad, n) for n, _, _ in self.layout}
        rows, grads, n = C.c_void_p(), C.c_void_p(), C.c_int32()
        check(lib.lstur_user_grad(self._ctx, C.byref(rows), C.byref(grads), C.byref(n)), 'lstur_user_grad')
        Ue = self.sizes.user_emb_dim
        if self.word_trainable:
            g['word_emb'] = self.word_grad.view(self.shapes['word_emb']).cpu().numpy()
        g['user_rows'] = self._peek_ptr(rows.value, n.value, torch.int32).cpu().numpy()
        g['user_row_grads'] = self._peek_ptr(grads.value, n.value * Ue, torch.float32).cpu().numpy().reshape(n.value, Ue)
        return g

    def _peek_ptr(self, ptr, count, dtype):
        esz = 4
        base = self.workspace.data_ptr()
        off = ptr - base
        return self.workspace[off:off + count * esz].view(dtype)

    def peek_device(self, name):
        """Device view (no copy, no sync) of an intermediate of the last call."""
        ptr, cnt = C.c_void_p(), C.c_size_t()
        check(lib.lstur_peek(self._ctx, name.encode(), C.byref(ptr), C.byref(cnt)), 'lstur_peek')
        return self._peek_ptr(ptr.value, cnt.value, torch.float32)

    def peek(self, name):
        ptr, cnt = C.c_void_p(), C.c_size_t()
        check(lib.lstur_peek(self._ctx, name.encode(), C.byref(ptr), C.byref(cnt)), 'lstur_peek')
        return self._peek_ptr(ptr.value, cnt.value, torch.float32).clone()

    # ---- weight interchange (reference utils.save_model / load_model, utils.py:65-79) -----------------
    def keras_weight_order(self):
        """Names in the order ``keras.Model.get_weights()`` lists them for the reference graph: trainable
        weights layer by layer (doc_encoder: Conv1D kernel/bias, attention kernel/bias, Dense kernel/bias;
        user_encoder: user Embedding, GRU kernel/recurrent/bias[, concat Dense]), then the frozen word table."""
        tok = ['qkv_w', 'qkv_b'] if self.news_encoder == 'mhsa' else ['conv_w', 'conv_b']
        names = tok + ['att_w', 'att_b', 'dense_w', 'dense_b', 'user_emb', 'gru_w', 'gru_u', 'gru_b']
        if self.arch in ('iigru', 'iicgru'):
            names += ['user_emb2']
        if self.arch in ('gru', 'iigru'):
            names += ['con_w', 'con_b']
        if self.arch == 'niatt':
            names += ['uatt_w', 'uatt_b']
        if self.score_model == 'ddot':
            names += ['sdu_w', 'sdu_b', 'sdd_w', 'sdd_b']
        elif self.score_model == 'dnn':
            names += ['sn1_w', 'sn1_b', 'sn2_w', 'sn2_b']
        return names + ['word_emb']

    def save(self, paths):
        """(json_path, pkl_path) like utils.save_model: a JSON description + a pickle of the weight list in
        Keras order and Keras shapes (attention kernel [F,1])."""
        import json
        json_path, pkl_path = paths
        w = self.get_weights()
        w['word_emb'] = self.word_emb.cpu().numpy()
        arrays = [w[n].reshape(-1, 1) if n in ('att_w', 'sn2_w', 'uatt_w') else w[n] for n in self.keras_weight_order()]
        with open(json_path, 'w') as f:
            json.dump(dict(format='mnexp_amd-lstur-1', arch=self.arch, order=self.keras_weight_order(),
                           shapes=[list(a.shape) for a in arrays]), f)
        with open(pkl_path, 'wb') as f:
            pickle.dump(arrays, f, protocol=pickle.HIGHEST_PROTOCOL)

    def load(self, paths):
        import json
        json_path, pkl_path = paths
        with open(json_path) as f:
            meta = json.load(f)
        if meta.get('arch') != self.arch:
            raise ValueError('saved model is arch {!r}, this model is {!r}'.format(meta.get('arch'), self.arch))
        arrays = load_weight_list(pkl_path)
        self.set_weights({n: np.asarray(a).reshape(self.shapes[n]) for n, a in zip(meta['order'], arrays)})

    @property
    def ENCODER_ORDER(self):
        tok = ['qkv_w', 'qkv_b'] if self.news_encoder == 'mhsa' else ['conv_w', 'conv_b']
        return ['word_emb'] + tok + ['att_w', 'att_b', 'dense_w', 'dense_b']

    def save_encoder(self, paths):
        """The news encoder alone, as ``utils.save_model(config.encoder_output, doc_encoder)`` would write it
        (utils.py:73-79): pickle of ``doc_encoder.get_weights()`` -- layer order, i.e. Embedding matrix, Conv1D
        kernel/bias, attention kernel [F,1]/bias, Dense kernel/bias."""
        import json
        json_path, pkl_path = paths
        w = self.get_weights()
        w['word_emb'] = self.word_emb.cpu().numpy()
        arrays = [w[n].reshape(-1, 1) if n == 'att_w' else w[n] for n in self.ENCODER_ORDER]
        with open(json_path, 'w') as f:
            json.dump(dict(format='mnexp_amd-lstur-encoder-1', order=self.ENCODER_ORDER,
                           shapes=[list(a.shape) for a in arrays]), f)
        with open(pkl_path, 'wb') as f:
            pickle.dump(arrays, f, protocol=pickle.HIGHEST_PROTOCOL)

    def load_encoder(self, paths):
        """``--enable-pretrain-encoder`` (paper.py:103-107: ``utils.load_model(config.encoder_input)``): takes the 7
        arrays of a saved cnnatt doc encoder (the list order above; the JSON side file is not interpreted)."""
        arrays = load_weight_list(paths[1])
        if len(arrays) != len(self.ENCODER_ORDER):
            raise ValueError('expected the {} weight arrays of a cnnatt doc encoder, found {}'.format(
                len(self.ENCODER_ORDER), len(arrays)))
        w = {}
        for n, a in zip(self.ENCODER_ORDER, arrays):
            a = np.asarray(a, dtype=np.float32)
            if a.size != int(np.prod(self.shapes[n])):
                raise ValueError('{}: saved shape {} does not fit {}'.format(n, a.shape, self.shapes[n]))
            w[n] = a.reshape(self.shapes[n])
        self.word_emb.copy_(torch.from_numpy(w.pop('word_emb')).to(self.device))
        self.set_weights(w)

    def profile(self, enable=True):
        """Bracket each kernel family with HIP events on the launch stream (bench.py roofline figures); enable = 2: the
        roofline kernels only (gather and the two conv contractions)."""
        check(lib.lstur_profile(self._ctx, int(enable)), 'lstur_profile')

    def profile_read(self):
        """{family: (total_ms, launches)}; synchronises on the recorded events."""
        out, i = {}, 0
        while True:
            name, ms, cnt = C.c_char_p(), C.c_float(), C.c_int32()
            rc = lib.lstur_profile_read(self._ctx, i, C.byref(name), C.byref(ms), C.byref(cnt))
            if rc == 1:
                break
            check(rc, 'lstur_profile_read')
            out[name.value.decode()] = (ms.value, cnt.value)
            i += 1
        return out

    def set_title_table(self, table):
        self.title_table = torch.from_numpy(np.ascontiguousarray(table, dtype=np.int32)).to(self.device)
        check(lib.lstur_set_title_table(self._ctx, self.title_table.data_ptr(), self.title_table.shape[0]),
              'lstur_set_title_table')

    # ---- device staging -------------------------------------------------------------
    def _dev_i32(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.int32)).to(self.device, non_blocking=True)

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def _split_reference_x(self, X, k1):
        """Reference X list -> (user, hist[n,W,T], cands[n,k1,T]) int32 device tensors.
        X = [user int64[n] (or [n,1]), clicked f64[n,W,T], k1 x cand f64[n,T]] (task/seq2vec.py:182-206)."""
        if not self.has_user:
            X = [np.zeros(len(X[0]), dtype=np.int32)] + list(X)
        if len(X) != 2 + k1:
            raise ValueError('expected {} input arrays, got {}'.format(2 + k1 - (not self.has_user), len(X) - (not self.has_user)))
        user = self._dev_i32(np.asarray(X[0]).reshape(-1))
        hist = self._dev_i32(X[1])
        cands = self._dev_i32(np.stack([np.asarray(x) for x in X[2:]], axis=1))
        return user, hist, cands

    # ---- training ---------------------------------------------------------------------
    def _dev_f32(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.float32)).to(self.device, non_blocking=True)

    def prefetch_ids(self, hist_ids, cand_ids, user=None):
        """Announce the NEXT training step's doc-id batch (device int32 tensors; kept alive here until that step): the
        library stages it under the current step's GRU recurrence (include/lstur.h: lstur_prefetch_ids).  user: that
        batch's user ids -- a lazily updated (large) user table brings those rows up to date there as well."""
        if self.pointwise:
            return
        self._prefetched = (hist_ids, cand_ids, user)
        if user is not None and self.has_user:
            check(lib.lstur_prefetch_rows(self._ctx, user.data_ptr()), 'lstur_prefetch_rows')
        check(lib.lstur_prefetch_ids(self._ctx, hist_ids.data_ptr(), cand_ids.data_ptr(), int(hist_ids.shape[0]),
                                     self.step + 1), 'lstur_prefetch_ids')

    def _train_step(self, user, hist, cands, ids_form, labels=None):
        n = int(user.shape[0])
        s = self._stream()
        if self.world == 1 and not self.force_collectives:
            # one process: forward + backward + Adam as ONE library call (include/lstur.h: lstur_train_step) -- the user-table
            # sweep runs under the encoder backward; same arithmetic as the split calls below
            fn = lib.lstur_train_step_ids if ids_form else lib.lstur_train_step
            t = self.optimizer.iterations + 1
            check(fn(self._ctx, user.data_ptr(), hist.data_ptr(), cands.data_ptr(),
                     labels.data_ptr() if self.pointwise else None, n, self.step, self.optimizer.lr, t,
                     self.stats.data_ptr(), s), 'lstur_train_step')
            if self.pointwise:
                self.auc.update(labels, self.peek_device('probs')[:n])
            self.optimizer.iterations = t
            self.step += 1
            return
        self._dp_prepare()
        if self.pointwise:
            fn = lib.lstur_point_forward_backward_ids if ids_form else lib.lstur_point_forward_backward
            check(fn(self._ctx, user.data_ptr(), hist.data_ptr(), cands.data_ptr(), labels.data_ptr(), n, self.step,
                     1.0 / self.world, self.stats.data_ptr(), s), 'lstur_point_forward_backward')
            self.auc.update(labels, self.peek_device('probs')[:n])
        else:
            fn = lib.lstur_forward_backward_ids if ids_form else lib.lstur_forward_backward
            check(fn(self._ctx, user.data_ptr(), hist.data_ptr(), cands.data_ptr(), n, self.step, 1.0 / self.world,
                     self.stats.data_ptr(), s), 'lstur_forward_backward')
        self.optimizer.iterations += 1
        t = self.optimizer.iterations
        if self.world > 1 or self.force_collectives:
            if self._dp_attached:
                # the backward pass staged this rank's rows into the bucket's tail; ONE all-reduce; the reduced tail goes to
                # the optimizer as it stands (include/lstur.h: lstur_dp_attach / lstur_adam_step_dp)
                from . import dist as D
                if self.word_trainable:
                    D.allreduce_bucket(self.word_grad, self.dist_group, prescaled=True)
                dense, used = self.sizes.dense, self._dp_used
                # (diagnostic, one rank only: LSTUR_DP_NO_COLLECTIVE=1 leaves the RCCL call out -- what is left of the split
                #  sequence's cost is the library's own share, tools/ab_dp.sh)
                if not (self.world == 1 and os.environ.get('LSTUR_DP_NO_COLLECTIVE')):
                    D.allreduce_bucket(self._bucket[:dense + used], self.dist_group, prescaled=True)
                check(lib.lstur_adam_step_dp(self._ctx, self.optimizer.lr, t, self._bucket[dense:].data_ptr(),
                                             max(1, self.world), n, s), 'lstur_adam_step_dp')
            else:
                rows, grads = self._allreduce_grads(n)
                check(lib.lstur_adam_step(self._ctx, self.optimizer.lr, t, rows.data_ptr(), grads.data_ptr(),
                                          int(rows.shape[0]), s), 'lstur_adam_step')
        else:
            check(lib.lstur_adam_step(self._ctx, self.optimizer.lr, t, None, None, 0, s), 'lstur_adam_step')
        self.step += 1

    def _dp_prepare(self):
        """the one-collective exchange without launches of its own (include/lstur.h: lstur_dp_attach): the backward pass
        stages this rank's user rows into the tail of the gradient bucket inside its own reduction launch.
        LSTUR_DP_FOLD=0: the stand-alone lstur_dp_stage / lstur_dp_unpack launches (same results)."""
        if self._dp_attached is not None:
            return
        from . import dist as D
        world = max(1, self.world)
        ok = (world <= 8 and self.cfg.n_users < (1 << 24) and self.sizes.dense % 4 == 0 and self.has_user
              and (self.world > 1 or self.force_collectives) and os.environ.get('LSTUR_DP_FOLD', '1') != '0')
        if ok:
            self._dp_used = int(lib.lstur_dp_tail_floats(C.byref(self.cfg), world))
            check(lib.lstur_dp_attach(self._ctx, self._bucket[self.sizes.dense:].data_ptr(), D.rank(self.dist_group), world),
                  'lstur_dp_attach')
        self._dp_attached = ok

    def _allreduce_grads(self, n):
        """Data parallelism over RCCL: ONE all-reduce per step.  The library stages this
        rank's (row, gradient) pairs into its slot of the tail behind the dense gradients (other slots zero), so the
        sum over ranks averages the dense bucket (already scaled by 1/world) AND concatenates the sparse rows."""
        from . import dist as D
        world = max(1, self.world)
        if self.word_trainable:          # V x D more gradient: its own all-reduce ("only if trainable")
            D.allreduce_bucket(self.word_grad, self.dist_group, prescaled=True)
        if world <= 8 and self.cfg.n_users < (1 << 24) and self.sizes.dense % 4 == 0:
            dense = self.sizes.dense
            used = int(lib.lstur_dp_tail_floats(C.byref(self.cfg), world))
            tail = self._bucket[dense:dense + used]
            rank = D.rank(self.dist_group)
            s = self._stream()
            check(lib.lstur_dp_stage(self._ctx, tail.data_ptr(), rank, world, s), 'lstur_dp_stage')
            D.allreduce_bucket(self._bucket[:dense + used], self.dist_group, prescaled=True)
            rows, grads = C.c_void_p(), C.c_void_p()
            check(lib.lstur_dp_unpack(self._ctx, tail.data_ptr(), world, n, C.byref(rows), C.byref(grads), s),
                  'lstur_dp_unpack')
            Ue = self.sizes.user_emb_dim
            return (self._peek_ptr(rows.value, world * n, torch.int32),
                    self._peek_ptr(grads.value, world * n * Ue, torch.float32).view(world * n, Ue))
        if self.arch in ('iigru', 'iicgru'):
            raise _lib.LsturError('iigru (two user tables): the data-parallel exchange needs the one-collective path '
                                  '(world <= 8, n_users < 2^24)')
        # general fallback: all-reduce of the bucket + all-gather of rows and gradients
        D.allreduce_bucket(self.dense_grad, self.dist_group, prescaled=True)
        rows, grads, cnt = C.c_void_p(), C.c_void_p(), C.c_int32()
        check(lib.lstur_user_grad(self._ctx, C.byref(rows), C.byref(grads), C.byref(cnt)), 'lstur_user_grad')
        Ue = self.sizes.user_emb_dim
        my_rows = self._peek_ptr(rows.value, n, torch.int32)
        my_grads = self._peek_ptr(grads.value, n * Ue, torch.float32).view(n, Ue)
        return D.allgather_rows(my_rows, my_grads, self.dist_group)

    def train_on_batch(self, X, Y=None):
        user, hist, cands = self._split_reference_x(X, self.K1)
        if self.pointwise:
            self._train_step(user, hist, cands[:, 0].contiguous(), False, self._dev_f32(np.asarray(Y).reshape(-1)))
        else:
            self._train_step(user, hist, cands, False)

    def _ids_to_dev(self, ids):
        if not self.has_user:
            ids = (np.zeros(len(ids[0]), dtype=np.int32),) + tuple(ids)
        user, hist_ids, cand_ids = ids
        return self._dev_i32(user), self._dev_i32(hist_ids), self._dev_i32(cand_ids)

    def train_on_ids(self, *ids, labels=None, next_ids=None, staged=None):
        """(user, hist_ids, cand_ids), or (hist_ids, cand_ids) for an id-less task; pointwise: + labels.
        next_ids: the batch of the FOLLOWING step, if known -- it is uploaded and announced to the library now, so its
        staging overlaps this step (returns its device tensors: pass them back as `staged` for that step)."""
        dev = staged if staged is not None else self._ids_to_dev(ids)
        nxt = None
        if next_ids is not None and not self.pointwise:
            nxt = self._ids_to_dev(next_ids)
            self.prefetch_ids(nxt[1], nxt[2], nxt[0])
        lab = self._dev_f32(np.asarray(labels).reshape(-1)) if self.pointwise else None
        self._train_step(dev[0], dev[1], dev[2], True, lab)
        return nxt

    def fit_generator(self, generator, steps_per_epoch, epochs=1, initial_epoch=0, verbose=2, ids_form=False,
                      workers=1, max_queue_size=10, **_ignored):
        """One ``train_on_batch`` per generator item (Keras fit_generator semantics, main.py:73-78).  Like Keras (defaults
        workers=1, max_queue_size=10: a GeneratorEnqueuer thread) the generator runs in ONE background thread, at most
        ``max_queue_size`` batches ahead -- and here exactly ``steps_per_epoch`` draws per epoch, so the RNG streams the
        generators consume advance as in a plain loop.  workers=0: the calling thread."""
        hist = History()
        source = generator
        for epoch in range(initial_epoch, epochs):
            self.stats_acc.zero_()
            generator = prefetched(source, steps_per_epoch, max_queue_size) if workers else source
            look = ids_form and not self.pointwise      # one batch of look-ahead: the next batch is staged under this step
            cur = next(generator) if look else None
            staged = None
            for i in range(steps_per_epoch):
                if look:
                    (X, Y), nxt = cur, None
                    if i + 1 < steps_per_epoch:
                        cur = next(generator)
                        nxt = cur[0]
                    staged = self.train_on_ids(*X, next_ids=nxt, staged=staged)
                    continue
                X, Y = next(generator)
                if ids_form:
                    self.train_on_ids(*X, labels=Y)
                else:
                    self.train_on_batch(X, Y)
            loss, acc = (self.stats_acc / steps_per_epoch).tolist()
            if self.pointwise:
                acc = self.auc.value()       # the running tf.metrics.auc value Keras logs for the epoch
            hist.history.setdefault('loss', []).append(loss)
            hist.history.setdefault(self.metrics_names[1], []).append(acc)
            if verbose:
                logging.info('Epoch %d/%d - loss: %.4f - %s: %.4f', epoch + 1, epochs, loss, self.metrics_names[1], acc)
        return hist

    def evaluate_generator(self, generator, steps, verbose=0, ids_form=False, **_ignored):
        acc = torch.zeros(2, dtype=torch.float32, device=self.device)
        total = 0
        for _ in range(steps):
            X, Y = next(generator)
            if ids_form:
                raise NotImplementedError('evaluate on id-form batches: expand with the title table first')
            user, hist, cands = self._split_reference_x(X, self.K1)
            n = int(user.shape[0])
            if self.pointwise:
                labels = self._dev_f32(np.asarray(Y).reshape(-1))
                check(lib.lstur_point_evaluate(self._ctx, user.data_ptr(), hist.data_ptr(),
                                               cands[:, 0].contiguous().data_ptr(), labels.data_ptr(), n,
                                               self.stats.data_ptr(), self._stream()), 'lstur_point_evaluate')
                self.auc.update(labels, self.peek_device('probs')[:n])
            else:
                check(lib.lstur_evaluate(self._ctx, user.data_ptr(), hist.data_ptr(), cands.data_ptr(), n,
                                         self.stats.data_ptr(), self._stream()), 'lstur_evaluate')
            acc += self.stats * n
            total += n
        out = (acc / max(total, 1)).tolist()
        if self.pointwise:
            out[1] = self.auc.value()
        return out

    # ---- inference ------------------------------------------------------------------------
    def _predict(self, user, hist, cand, ids_form):
        n = int(user.shape[0])
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        cap = self.sizes.max_predict_rows
        fn = lib.lstur_predict_ids if ids_form else lib.lstur_predict
        for s0 in range(0, n, cap):
            s1 = min(n, s0 + cap)
            check(fn(self._ctx, user[s0:s1].data_ptr(), hist[s0:s1].contiguous().data_ptr(),
                     cand[s0:s1].contiguous().data_ptr(), s1 - s0, out[s0:s1].data_ptr(), self._stream()),
                  'lstur_predict')
        return out.cpu().numpy().reshape(-1, 1)

    def predict(self, X, **_ignored):
        """test_model.predict: X = [user[n], clicked[n,W,T], candidate[n,T]] -> [n,1] sigmoid(u.d)."""
        user, hist, cands = self._split_reference_x(X, 1)
        return self._predict(user, hist, cands[:, 0].contiguous(), False)

    predict_on_batch = predict

    def predict_ids(self, *ids):
        if not self.has_user:
            ids = (np.zeros(len(ids[0]), dtype=np.int32),) + tuple(ids)
        user, hist_ids, cand_ids = ids
        c = self._dev_i32(cand_ids).reshape(len(user), -1)[:, 0].contiguous()
        return self._predict(self._dev_i32(user), self._dev_i32(hist_ids), c, True)

    def __del__(self):
        try:
            if getattr(self, '_ctx', None):
                lib.lstur_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass
