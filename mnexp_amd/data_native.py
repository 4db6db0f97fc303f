"""ctypes binding of liblstur_data.so (include/lstur_data.h): the native training-sample feed.

`NativeTrainFeed(task)` yields the same id-form batches as ``task.train_ids`` -- bit for bit for the same
seed (the NumPy legacy RNG stream is restated in C++) -- without the per-sample Python work and the
per-batch list shuffle of 100*B tuples that cap the reference's generator at ~20 k samples/s.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblstur_data.so')
_P, _I = C.c_void_p, C.c_int32

SIGNATURES = {
    'lstur_data_last_error': (C.c_char_p, []),
    'lstur_data_create': (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(_P)]),
    'lstur_data_destroy': (_I, [_P]),
    'lstur_data_seed': (_I, [_P, C.c_uint32]),
    'lstur_data_next_train': (_I, [_P, _P, _P, _P]),
    'lstur_data_next_samples': (_I, [_P, _I, _P, _P, _P]),
    'lstur_data_peek_rng': (_I, [_P, _P]),
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


def _check(rc, what):
    if rc != 0:
        raise RuntimeError('{} failed: {}'.format(what, lib.lstur_data_last_error().decode()))


class NativeTrainFeed:
    """Endless iterator of ((user int64[B], hist int32[B,W], cands int32[B,K+1]), labels int64[B,K+1]); the id-less
    Seq2VecPaperSoftmax gets (hist, cands) -- the shape its own ``train_ids`` yields.  Users are sharded by rank exactly
    like ``Seq2VecPaper._users(shard=True)``."""

    def __init__(self, task, seed=None, rank=None, world=None):
        from . import dist
        cfg = task.config
        if getattr(task, 'head', 'softmax') != 'softmax' or type(task).__name__.find('Days') >= 0:
            raise NotImplementedError('--native-feed covers the Seq2VecPaperSoftmax(Id) sample stream only; use --ids '
                                      'for {}'.format(type(task).__name__))
        self.B, self.W, self.K = cfg.batch_size, cfg.window_size, cfg.negative_samples
        self.has_user = bool(getattr(task, 'has_user', True))      # id-less task: batches carry no user column
        rank = dist.rank() if rank is None else rank
        world = dist.world_size() if world is None else world
        imp_off, pos_off, neg_off, pos, neg = [0], [0], [0], [], []
        for train, _ in task.data:
            for imp in train:
                pos.extend(imp.pos)
                neg.extend(imp.neg)
                pos_off.append(len(pos))
                neg_off.append(len(neg))
            imp_off.append(len(pos_off) - 1)
        a = lambda x, t: np.ascontiguousarray(np.asarray(x, dtype=t))
        self._arrays = (a(imp_off, np.int64), a(pos_off, np.int64), a(pos, np.int32), a(neg_off, np.int64), a(neg, np.int32))
        io, po, p, no, n = self._arrays
        self._h = C.c_void_p()
        _check(lib.lstur_data_create(len(task.data), io.ctypes.data, po.ctypes.data, p.ctypes.data, no.ctypes.data,
                                     n.ctypes.data, self.W, self.K, self.B, rank, world, C.byref(self._h)),
               'lstur_data_create')
        self.labels = np.tile(np.array([1] + [0] * self.K, dtype=np.int64), (self.B, 1))
        if seed is not None:
            self.seed(seed)

    def seed(self, seed):
        """np.random.seed(seed) for the feed's private generator; restarts the walk."""
        _check(lib.lstur_data_seed(self._h, int(seed) & 0xFFFFFFFF), 'lstur_data_seed')

    def __iter__(self):
        return self

    def __next__(self):
        user = np.empty(self.B, np.int32)
        hist = np.empty((self.B, self.W), np.int32)
        cands = np.empty((self.B, self.K + 1), np.int32)
        _check(lib.lstur_data_next_train(self._h, user.ctypes.data, hist.ctypes.data, cands.ctypes.data), 'next_train')
        if not self.has_user:
            return (hist, cands), self.labels
        return (user.astype(np.int64), hist, cands), self.labels

    def samples(self, n):
        user = np.empty(n, np.int32)
        hist = np.empty((n, self.W), np.int32)
        cands = np.empty((n, self.K + 1), np.int32)
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
