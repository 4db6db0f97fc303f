"""ctypes binding of liblstur_data.so (include/lstur_data.h): native loaders and sample feeds of the LSTUR path.

* ``load_docs`` / ``ClickData`` / ``load_vocab``: DocMeta.tsv, ClickData.tsv and Vocab.tsv parsed in C++ -- the same
  tables, the same failures as the Python mirror (``task/seq2vec.py``, ``task/paper.py``, ``utils.py``: pinned by
  ``tests/golden`` against the reference itself), seconds instead of minutes at 1 M users / 200 k docs.
* ``NativeFeed(task)``: the task family's train / valid / test generators over the loaded ClickData in id form -- bit
  for bit the batches of ``task.train_ids`` / ``task.valid_ids`` / ``task.test_gen_ids`` for the same seed (the NumPy
  legacy RNG stream is restated in C++) -- without the per-sample Python work and the per-batch list shuffle of 100*B
  tuples that cap the reference's generator at ~20 k samples/s.  Listwise (Seq2VecPaperSoftmax[Id]), pointwise
  (Seq2VecPaper / ...Dot / ...Id) and the time-window `...Days` variants.
"""
import ctypes as C
import os
from datetime import datetime, timedelta

import random

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblstur_data.so')
_P, _I, _L = C.c_void_p, C.c_int32, C.c_int64

SIGNATURES = {
    'lstur_data_last_error': (C.c_char_p, []),
    'lstur_data_create': (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    'lstur_data_create_from_clicks': (_I, [_P, _I, _I, _I, _I, _I, _I, C.c_double, C.POINTER(_P)]),
    'lstur_data_destroy': (_I, [_P]),
    'lstur_data_seed': (_I, [_P, C.c_uint32]),
    'lstur_data_pyrandom': (_I, [_P, _P, _I, _I]),
    'lstur_data_next_train': (_I, [_P, _P, _P, _P]),
    'lstur_data_next_train_rows': (_I, [_P, _P, _P, _P, _P]),
    'lstur_data_next_valid': (_I, [_P, _P, _P, _P, _P]),
    'lstur_data_test_reset': (_I, [_P]),
    'lstur_data_test_rows': (_L, [_P]),
    'lstur_data_test_next': (_I, [_P, _P, _P, _P, _P]),
    'lstur_data_next_samples': (_I, [_P, _I, _P, _P, _P]),
    'lstur_data_peek_rng': (_I, [_P, _P]),
    'lstur_docs_load': (_I, [C.c_char_p, _I, C.POINTER(_P)]),
    'lstur_docs_count': (_I, [_P]),
    'lstur_docs_titles': (_P, [_P]),
    'lstur_docs_present': (_P, [_P]),
    'lstur_docs_destroy': (_I, [_P]),
    'lstur_clicks_load': (_I, [C.c_char_p, _P, _I, C.POINTER(_P)]),
    'lstur_clicks_users': (_I, [_P]),
    'lstur_clicks_timed': (_I, [_P]),
    'lstur_clicks_arrays': (_I, [_P, _I] + [C.POINTER(_P)] * 6 + [C.POINTER(_L)] * 3),
    'lstur_clicks_destroy': (_I, [_P]),
    'lstur_vocab_load': (_I, [C.c_char_p, _I, C.POINTER(_P), C.POINTER(_I)]),
    'lstur_vocab_free': (None, [_P]),
    'lstur_synth_write': (_I, [C.c_char_p, C.c_char_p, _I, _I, _I, _I, C.c_double, C.c_double, C.c_uint64]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('liblstur_data.so is not built ({}); run `make -C mnexp_amd/csrc`'.format(LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class NativeDataError(ValueError):
    """what the reference raises as ValueError / IndexError / KeyError while parsing, with the file and line"""


def _check(rc, what):
    if rc != 0:
        raise NativeDataError('{} failed: {}'.format(what, lib.lstur_data_last_error().decode()))


def _view(ptr, n, dtype):
    """numpy view of n elements at a C pointer (no copy; the owner must outlive it)"""
    if n == 0:
        return np.zeros(0, dtype)
    ct = {np.int32: C.c_int32, np.int64: C.c_int64, np.uint8: C.c_uint8, np.float32: C.c_float}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))


def synth_write(out_dir, n_docs, vocab, title_len, n_users, lam_train=20, lam_valid=4, seed=0):
    """DocMeta.tsv + ClickData.tsv of a workload's shape into out_dir (the C++ writer: a 1 M-user set in a minute)"""
    _check(lib.lstur_synth_write(os.fsencode(os.path.join(out_dir, 'DocMeta.tsv')), os.fsencode(os.path.join(out_dir, 'ClickData.tsv')),
                                 int(n_docs), int(vocab), int(title_len), int(n_users), float(lam_train), float(lam_valid), int(seed)),
           'lstur_synth_write')


# ---- loaders -------------------------------------------------------------------------------------------------------
def load_docs(path, title_len):
    """DocMeta.tsv -> (int32 [doc_count, T] title table, bool [doc_count] present); Seq2Vec._load_docs natively."""
    h = C.c_void_p()
    _check(lib.lstur_docs_load(os.fsencode(path), int(title_len), C.byref(h)), 'lstur_docs_load')
    try:
        n = lib.lstur_docs_count(h)
        table = _view(lib.lstur_docs_titles(h), n * title_len, np.int32).reshape(n, title_len).copy()
        present = _view(lib.lstur_docs_present(h), n, np.uint8).astype(bool)
    finally:
        lib.lstur_docs_destroy(h)
    return table, present


def load_vocab(path, dim):
    """Vocab.tsv -> float32 [max_index + 1, dim]; utils.load_textual_embedding natively."""
    m, rows = C.c_void_p(), C.c_int32()
    _check(lib.lstur_vocab_load(os.fsencode(path), int(dim), C.byref(m), C.byref(rows)), 'lstur_vocab_load')
    try:
        return _view(m, rows.value * dim, np.float32).reshape(rows.value, dim).copy()
    finally:
        lib.lstur_vocab_free(m)


class ClickData:
    """ClickData.tsv in CSR form (Seq2VecPaper._load_data natively).  ``train`` / ``valid``: dicts of numpy VIEWS
    (imp_off, pos_off, pos, neg_off, neg, time) into the native handle, which this object owns."""

    def __init__(self, path, present=None):
        self._h = C.c_void_p()
        pp = np.ascontiguousarray(present, dtype=np.uint8) if present is not None else None
        _check(lib.lstur_clicks_load(os.fsencode(path), pp.ctypes.data if pp is not None else None,
                                     len(pp) if pp is not None else 0, C.byref(self._h)), 'lstur_clicks_load')
        self.n_users = lib.lstur_clicks_users(self._h)
        self.timed = bool(lib.lstur_clicks_timed(self._h))
        self.train, self.valid = self._arrays(0), self._arrays(1)

    def _arrays(self, which):
        ptr = [C.c_void_p() for _ in range(6)]
        cnt = [C.c_int64() for _ in range(3)]
        _check(lib.lstur_clicks_arrays(self._h, which, *[C.byref(p) for p in ptr], *[C.byref(c) for c in cnt]), 'lstur_clicks_arrays')
        n_imp, n_pos, n_neg = (c.value for c in cnt)
        return dict(imp_off=_view(ptr[0], self.n_users + 1, np.int64), pos_off=_view(ptr[1], n_imp + 1, np.int64),
                    pos=_view(ptr[2], n_pos, np.int32), neg_off=_view(ptr[3], n_imp + 1, np.int64),
                    neg=_view(ptr[4], n_neg, np.int32), time=_view(ptr[5], n_imp, np.int64))

    def __len__(self):
        return self.n_users

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                lib.lstur_clicks_destroy(self._h)
                self._h = None
        except Exception:
            pass


_EPOCH = datetime(1970, 1, 1)


class LazyUsers:
    """``task.data`` over a native ClickData: ``data[u] -> (training impressions, validation impressions)`` as the task's
    own Impression objects, built on demand -- the Python generators of the task (the reference protocol) keep working
    on a natively loaded set without a million-row list of Python objects."""

    def __init__(self, clicks, impression_cls):
        self.clicks, self.cls = clicks, impression_cls
        self.has_time = 'time' in getattr(impression_cls, '__slots__', ())

    def __len__(self):
        return self.clicks.n_users

    def _imps(self, a, u):
        out = []
        for i in range(int(a['imp_off'][u]), int(a['imp_off'][u + 1])):
            imp = self.cls.__new__(self.cls)
            imp.pos = a['pos'][a['pos_off'][i]:a['pos_off'][i + 1]].tolist()
            imp.neg = a['neg'][a['neg_off'][i]:a['neg_off'][i + 1]].tolist()
            if self.has_time and a['time'][i] != np.iinfo(np.int64).min:
                imp.time = _EPOCH + timedelta(seconds=int(a['time'][i]))
            out.append(imp)
        return out

    def __getitem__(self, u):
        if u < 0:
            u += len(self)
        if not 0 <= u < len(self):
            raise IndexError(u)
        return self._imps(self.clicks.train, u), self._imps(self.clicks.valid, u)

    def __iter__(self):
        return (self[u] for u in range(len(self)))


# ---- feeds ---------------------------------------------------------------------------------------------------------
class NativeFeed:
    """The task's sample generators natively.  Iterating yields TRAINING batches (pool-shuffled, the shape of
    ``task.train_ids``): listwise ``((user int64[B], hist int32[B,W], cands int32[B,K+1]), labels int64[B,K+1])``, pointwise
    ``((user, hist, cand int32[B,1]), labels int64[B])``; the id-less tasks get no user column.  ``valid_batches()`` and
    ``test_impressions()`` are the other two generators.  Users are sharded by rank for TRAINING exactly like
    ``Seq2VecPaper._users(shard=True)``; validation / test walk every user."""

    def __init__(self, task, seed=None, rank=None, world=None):
        from . import dist
        cfg = task.config
        head = getattr(task, 'head', 'softmax')
        self.pointwise = head != 'softmax'
        self.days = float(cfg.days) if type(task).__name__.find('Days') >= 0 else 0.0
        self.B, self.W, self.K = cfg.batch_size, cfg.window_size, cfg.negative_samples
        self.C = 1 if self.pointwise else self.K + 1
        self.has_user = bool(getattr(task, 'has_user', True))      # id-less task: batches carry no user column
        rank = dist.rank() if rank is None else rank
        world = dist.world_size() if world is None else world
        self._h = C.c_void_p()
        clicks = getattr(task, 'clicks', None)
        if clicks is None:      # a Python-loaded task: CSR of the training histories from its objects (listwise only)
            if self.pointwise or self.days:
                raise NotImplementedError('--native-feed for {} needs the natively loaded ClickData (--native-load)'.format(
                    type(task).__name__))
            imp_off, pos_off, neg_off, pos, neg = [0], [0], [0], [], []
            for train, _ in task.data:
                for imp in train:
                    pos.extend(imp.pos)
                    neg.extend(imp.neg)
                    pos_off.append(len(pos))
                    neg_off.append(len(neg))
                imp_off.append(len(pos_off) - 1)
            a = lambda x, t: np.ascontiguousarray(np.asarray(x, dtype=t))
            io, po, p, no, n = (a(imp_off, np.int64), a(pos_off, np.int64), a(pos, np.int32), a(neg_off, np.int64), a(neg, np.int32))
            _check(lib.lstur_data_create(len(task.data), io.ctypes.data, po.ctypes.data, p.ctypes.data, no.ctypes.data,
                                         n.ctypes.data, self.W, self.K, self.B, rank, world, C.byref(self._h)),
                   'lstur_data_create')
        else:
            _check(lib.lstur_data_create_from_clicks(clicks._h, self.W, self.K, self.B, rank, world, int(self.pointwise),
                                                     self.days, C.byref(self._h)), 'lstur_data_create_from_clicks')
        self.labels = np.tile(np.array([1] + [0] * self.K, dtype=np.int64), (self.B, 1))
        # pointwise tasks sub-sample users with more than --max-impression training impressions through Python's `random`
        # (task/paper.py:271-272): the feed continues from the module's current state, privately
        self._max_impression = int(cfg.max_impression) if self.pointwise else 0
        self._pyrandom(random.getstate())
        if seed is not None:
            self.seed(seed)

    def _pyrandom(self, state):
        if not self._max_impression:
            return
        words = np.asarray(state[1], dtype=np.uint32)          # 624 state words + the position
        _check(lib.lstur_data_pyrandom(self._h, words[:624].ctypes.data, int(words[624]), self._max_impression),
               'lstur_data_pyrandom')

    def seed(self, seed):
        """np.random.seed(seed) -- and random.seed(seed) -- for the feed's private generators; restarts every walk."""
        self._pyrandom(random.Random(int(seed)).getstate())
        _check(lib.lstur_data_seed(self._h, int(seed) & 0xFFFFFFFF), 'lstur_data_seed')

    def _pack(self, user, hist, cands, label):
        ids = (user.astype(np.int64), hist, cands) if self.has_user else (hist, cands)
        return ids, (label.astype(np.int64) if self.pointwise else self.labels)

    def _batch(self, fn, what):
        user, hist = np.empty(self.B, np.int32), np.empty((self.B, self.W), np.int32)
        cands, label = np.empty((self.B, self.C), np.int32), np.empty(self.B, np.int32)
        _check(fn(self._h, user.ctypes.data, hist.ctypes.data, cands.ctypes.data, label.ctypes.data), what)
        return self._pack(user, hist, cands, label)

    def __iter__(self):
        return self

    def __next__(self):
        return self._batch(lib.lstur_data_next_train_rows, 'next_train')

    def valid_batches(self):
        """task.valid_ids natively: endless, unshuffled batches of B validation samples"""
        while True:
            yield self._batch(lib.lstur_data_next_valid, 'next_valid')

    def test_impressions(self, rank=0, world=1):
        """task.test_gen_ids natively: per validation impression ``((user, hist, cand), labels)`` -- all positives, then
        all negatives, the history frozen in front of the impression.  world > 1: only impressions rank, rank + world, ...
        (data-parallel evaluation; yields ``(index, batch)`` pairs then)."""
        _check(lib.lstur_data_test_reset(self._h), 'test_reset')
        i = 0
        while True:
            n = lib.lstur_data_test_rows(self._h)
            if n <= 0:
                return
            if world > 1 and i % world != rank:
                _check(lib.lstur_data_test_next(self._h, None, None, None, None), 'test_skip')
                i += 1
                continue
            user, hist = np.empty(n, np.int32), np.empty((n, self.W), np.int32)
            cand, label = np.empty((n, 1), np.int32), np.empty(n, np.int32)
            _check(lib.lstur_data_test_next(self._h, user.ctypes.data, hist.ctypes.data, cand.ctypes.data, label.ctypes.data),
                   'test_next')
            ids = (user.astype(np.int64), hist, cand) if self.has_user else (hist, cand)
            yield ((i, (ids, label.astype(np.int64))) if world > 1 else (ids, label.astype(np.int64)))
            i += 1

    def samples(self, n):
        user = np.empty(n, np.int32)
        hist = np.empty((n, self.W), np.int32)
        cands = np.empty((n, self.C), np.int32)
        _check(lib.lstur_data_next_samples(self._h, n, user.ctypes.data, hist.ctypes.data, cands.ctypes.data), 'next_samples')
        return user, hist, cands

    def peek_rng(self):
        out = np.empty(4, np.int32)
        _check(lib.lstur_data_peek_rng(self._h, out.ctypes.data), 'peek_rng')
        return out

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                lib.lstur_data_destroy(self._h)
                self._h = None
        except Exception:
            pass


NativeTrainFeed = NativeFeed      # (the name of round 1-2: the training generator alone)
