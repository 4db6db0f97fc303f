"""LSTUR task family (softmax ranking with user-id embedding) on the native MI355X model.

Host-side mirror of /root/reference/task/paper.py:
  ``Seq2VecPaper``          :6-255   in-memory ClickData (``_load_data`` :24-35, ``Impression`` :7-18); itself the
                                     pointwise task: one candidate per sample (1 positive, then K negatives as
                                     separate rows :38-60), concat-MLP scorer :223-227, ``Seq2Vec.loss``
  ``Seq2VecPaperDot``       :258-262 the same with sigmoid(u . d)
  ``Seq2VecPaperId``        :265-383 the same with the user index (= ClickData row) as first input
  ``Seq2VecPaperSoftmax``   :384-524 1+K softmax samples, lr decay + per-impression metrics ``callback`` :497-524
  ``Seq2VecPaperSoftmaxId`` :527-665 user index = ClickData row number; LSTUR-ini (``igru``) / LSTUR-con (``gru``, ``hgru``)
Everything under ``_build_model`` is replaced by ``mnexp_amd.native_model.NativeModel`` (HIP/gfx950).
"""
import logging
import os
import random
from datetime import datetime, timedelta

import numpy as np
from sklearn.metrics import roc_auc_score

from .. import utils
from .seq2vec import Seq2Vec


class Seq2VecPaper(Seq2Vec):
    class Impression:
        __slots__ = ['pos', 'neg', 'time']

        def __init__(self, d):
            fields = d.split('#TAB#')
            self.pos = [int(k) for k in fields[0].split(' ')]
            self.neg = [int(k) for k in fields[1].split(' ')]
            if len(fields) >= 3:
                self.time = datetime.strptime(fields[2], '%m/%d/%Y %I:%M:%S %p')

        def negative_samples(self, n):
            return np.random.choice(self.neg, n)

    def _extract_impressions(self, x):
        # impressions without positives (segment starts with '#TAB#') are dropped (paper.py:21)
        return [self.Impression(d) for d in x.split('#N#') if not d.startswith('#TAB#')]

    def _load_data(self):
        """ClickData.tsv col[2] = training-period history, col[3] = validation-period history."""
        logging.info('[+] loading data')
        self._feed = None
        if getattr(self.config, 'native_load', False):
            # C++ loader (include/lstur_data.h): CSR arrays instead of a list of Python objects per user; the doc-id check
            # runs inside it; `self.data` stays a sequence of (training impressions, validation impressions)
            from ..data_native import ClickData, LazyUsers
            self.clicks = ClickData(self.config.training_data_input, present=self.docs.present)
            self.data = LazyUsers(self.clicks, self.Impression)
            super(Seq2VecPaper, self)._load_data()
            logging.info('[-] loaded data')
            return
        self.clicks = None
        self.data = []
        with open(self.config.training_data_input) as file:
            for line in file:
                cols = line.strip('\n').split('\t')
                self.data.append((
                    self._extract_impressions(cols[2]) if cols[2] else [],
                    self._extract_impressions(cols[3]) if cols[3] else [],
                ))
        for tr, va in self.data:                      # fail at load time on unknown doc ids
            for imp in tr + va:
                self.docs.check(imp.pos)
                self.docs.check(imp.neg)
        super(Seq2VecPaper, self)._load_data()
        logging.info('[-] loaded data')

    def save_model(self):
        pass                                          # paper.py:254-255

    def native_feed(self):
        """ONE C++ feed per task (training, validation and test generators share its generator state like the reference's
        share NumPy's global one) when --native-feed is set and the configuration is on its path; else None"""
        if not getattr(self.config, 'native_feed', False):
            return None
        if getattr(self, '_feed', None) is None:
            from ..data_native import NativeFeed
            try:
                self._feed = NativeFeed(self, seed=int(getattr(self.config, 'seed', 0) or 0))
            except NotImplementedError as e:
                logging.info('[*] native feed not used: {}'.format(e))
                self._feed = False
        return self._feed or None

    # ---- pointwise samples: (user?, hist_ids[W], [candidate], label) -------------------------------
    has_user = False
    head = 'point_mlp'                                # score_encoder, paper.py:223-227

    def _users(self, shard=False):
        """(user id or None, (train impressions, valid impressions)) per ClickData row.  ``shard=True`` (the TRAINING
        generators under data parallelism): only the rows this rank owns, r, r+N, ... (mnexp_amd/dist.py)
        -- for the id-less tasks too, or every rank would train on the same samples.  The validation generator walks every
        user on every rank; the test impressions of the callback are dealt out over the ranks (`_report`), which all-reduces
        the metric sums: the reported numbers are those of the single-GPU run."""
        from .. import dist
        w, r = dist.world_size(), dist.rank()
        idx = dist.shard_users(len(self.data), r, w) if (shard and w > 1) else range(len(self.data))
        return (((u if self.has_user else None), self.data[u]) for u in idx)

    def _no_samples(self):
        from .. import dist
        return RuntimeError('rank {} of {}: no user of this rank\'s shard yields a training sample (every rank must feed '
                            'the same number of steps into the all-reduce)'.format(dist.rank(), dist.world_size()))

    def _row(self, user, hist, cand, label):
        head = (user,) if self.has_user else ()
        return head + (hist, [int(cand)], label)

    def train_gen_ids(self):
        """paper.py:38-60 / :266-288: per positive click with a non-empty history one positive row, then K rows of
        sampled negatives; users with more than ``max_impression`` training impressions train on a random subset
        (``random.sample``) while every click still enters the history."""
        K = self.config.negative_samples
        while True:
            emitted = False
            for user, (ih, _) in self._users(shard=True):
                if not ih:
                    continue
                ch = self.Window(self.docs, self.config.window_size)
                if len(ih) > self.config.max_impression:
                    chosen = set(random.sample(ih, self.config.max_impression))
                else:
                    chosen = None
                for impression in ih:
                    trainable = chosen is None or impression in chosen
                    for pos in impression.pos:
                        if ch.count and trainable:
                            hist = ch.get_ids()
                            emitted = True
                            yield self._row(user, hist, pos, 1)
                            for neg in impression.negative_samples(K):
                                yield self._row(user, hist, neg, 0)
                        ch.push(pos)
            if not emitted:
                raise self._no_samples()

    def valid_gen_ids(self):
        """paper.py:62-77 / :290-305."""
        K = self.config.negative_samples
        while True:
            for user, (ih1, ih2) in self._users():
                if not (ih1 and ih2):
                    continue
                ch = self._history_after_training(ih1)
                for impression in ih2:
                    for pos in impression.pos:
                        hist = ch.get_ids()
                        yield self._row(user, hist, pos, 1)
                        for neg in impression.negative_samples(K):
                            yield self._row(user, hist, neg, 0)
                    for pos in impression.pos:
                        ch.push(pos)

    def test_gen_ids(self):
        """paper.py:79-97 / :307-328: per valid impression all pos then all neg, history frozen before it."""
        for user, (ih1, ih2) in self._users():
            if not (ih1 and ih2):
                continue
            ch = self._history_after_training(ih1)
            for impression in ih2:
                hist = ch.get_ids()
                yield ([self._row(user, hist, p, 1) for p in impression.pos] +
                       [self._row(user, hist, n, 0) for n in impression.neg])
                for pos in impression.pos:
                    ch.push(pos)

    def _history_after_training(self, ih1):
        ch = self.Window(self.docs, self.config.window_size)
        for impression in ih1:
            for pos in impression.pos:
                ch.push(pos)
        return ch

    # ---- batching ------------------------------------------------------------
    def _stack_ids(self, samples):
        cols = list(zip(*samples))
        labels = np.stack(cols[-1])
        hist = np.stack(cols[-3]).astype(np.int32)
        cands = np.array(cols[-2], dtype=np.int32)
        if self.has_user:
            return (np.array(cols[0], dtype=np.int64), hist, cands), labels
        return (hist, cands), labels

    def _expand(self, samples):
        ids, labels = self._stack_ids(samples)
        hist, cands = ids[-2], ids[-1]
        tf64 = self.docs.titles_f64
        X = [tf64[hist]] + [tf64[cands[:, j]] for j in range(cands.shape[1])]
        if self.has_user:
            X = [ids[0]] + X
        return X, labels

    # ---- model -----------------------------------------------------------------
    def _build_model(self):
        from ..native_model import NativeModel
        self.model = NativeModel.from_task(self, user_count=len(self.data) if self.has_user else 0)
        self.test_model = self.model.test_view()

    def _report(self, limit):
        """Mean per-impression AUC / nDCG@10 / nDCG@5 / MRR over the first `limit` validation impressions (paper.py:504-516).
        Under data parallelism the impressions are dealt out over the ranks (i % world == rank) and the per-metric sums and
        the count are all-reduced: every rank reports the single-process means without predicting every impression."""
        from .. import dist
        w, r = dist.world_size(), dist.rank()
        sums, n = np.zeros(7), 0
        for i, y_pred, y_true in self.test_items(limit, r, w):
            sums += (roc_auc_score(y_true, y_pred), utils.ndcg_score(y_true, y_pred, 10), utils.ndcg_score(y_true, y_pred, 5),
                     utils.mrr_score(y_true, y_pred), np.sum(y_true), len(y_true), i)
            n += 1
        if w > 1:
            tot = dist.allreduce_sums(list(sums) + [n])
            sums, n = np.array(tot[:7]), int(round(tot[7]))
        v = sums / max(n, 1)
        utils.logging_evaluation(dict(auc=v[0], ndcgx=v[1], ndcgv=v[2], mrr=v[3]))
        utils.logging_evaluation(dict(pos=v[4], size=v[5], num=v[6] * 2 + 1))
        return dict(auc=v[0], ndcgx=v[1], ndcgv=v[2], mrr=v[3])

    def callback(self, epoch):
        """task/seq2vec.py:296-321 (the pointwise tasks do not override it): the running AUC metric is
        re-initialised, lr *= decay, then mean per-impression AUC / nDCG@10 / nDCG@5 / MRR over the first
        ``testing_impression`` impressions of ``self.test``; repeated after the last epoch when TestData.tsv exists."""
        if getattr(self.model, 'auc', None) is not None:
            self.model.auc.reset()
        self.model.optimizer.lr = self.model.optimizer.lr * self.config.learning_rate_decay

        def report():
            return self._report(self.config.testing_impression)

        self.last_metrics = report()
        if epoch == self.config.epochs - 1 and os.path.exists(self.config.testing_data_input):
            self.is_training = False
            self.last_metrics = report()


class Seq2VecPaperDot(Seq2VecPaper):
    head = 'point_dot'                                # paper.py:258-262


class Seq2VecPaperId(Seq2VecPaper):
    """User id = row number of ClickData.tsv (paper.py:268)."""
    has_user = True


class Seq2VecPaperSoftmax(Seq2VecPaper):
    """Samples: (hist_ids[W], [pos, neg_1..neg_K], label=[1,0..0])."""
    has_user = False
    head = 'softmax'

    def _label(self):
        return [1] + [0] * self.config.negative_samples

    def _sample(self, user, ch, pos, impression, label):
        cands = [pos] + [int(n) for n in impression.negative_samples(self.config.negative_samples)]
        head = (user,) if self.has_user else ()
        return head + (ch.get_ids(), cands, label)

    def train_gen_ids(self):
        """paper.py:385-399 / :528-542: history excludes the current click and grows inside an impression."""
        label = self._label()
        while True:
            emitted = False
            for user, (ih, _) in self._users(shard=True):
                if not ih:
                    continue
                ch = self.Window(self.docs, self.config.window_size)
                for impression in ih:
                    for pos in impression.pos:
                        if ch.count:
                            emitted = True
                            yield self._sample(user, ch, pos, impression, label)
                        ch.push(pos)
            if not emitted:
                raise self._no_samples()

    def valid_gen_ids(self):
        """paper.py:401-418 / :544-562: history = all training clicks; pushed after each valid impression."""
        label = self._label()
        while True:
            for user, (ih1, ih2) in self._users():
                if not (ih1 and ih2):
                    continue
                ch = self._history_after_training(ih1)
                for impression in ih2:
                    for pos in impression.pos:
                        yield self._sample(user, ch, pos, impression, label)
                    for pos in impression.pos:
                        ch.push(pos)

    def test_gen_ids(self):
        """paper.py:420-441 / :563-582: per valid impression all pos then all neg, history frozen before it."""
        for user, (ih1, ih2) in self._users():
            if not (ih1 and ih2):
                continue
            ch = self._history_after_training(ih1)
            for impression in ih2:
                hist = ch.get_ids()
                head = (user,) if self.has_user else ()
                yield ([head + (hist, [p], 1) for p in impression.pos] +
                       [head + (hist, [n], 0) for n in impression.neg])
                for pos in impression.pos:
                    ch.push(pos)

    def callback(self, epoch):
        """paper.py:497-524: lr *= decay every epoch; mean per-impression AUC / nDCG@10 / nDCG@5 / MRR."""
        self.model.optimizer.lr = self.model.optimizer.lr * self.config.learning_rate_decay
        self.model, self.test_model = self.test_model, self.model

        def report(limit):
            return self._report(limit)

        self.last_metrics = report(self.config.validation_impression)
        if epoch == self.config.epochs - 1:
            self.is_training = False
            self.last_metrics = report(self.config.testing_impression)
        self.model, self.test_model = self.test_model, self.model


class Seq2VecPaperSoftmaxId(Seq2VecPaperSoftmax):
    """User id = row number of ClickData.tsv (paper.py:531)."""
    has_user = True


class _ExpiringWindow:
    """Time-aware click window of the `...Days` tasks (paper.py:669-696, 763-792): an entry is visible until
    `days` after its click; an expired entry reads as the pad doc 0 IN PLACE (the history keeps its holes)."""
    __slots__ = ['docs', 'click_history', 'click_history_time', 'delta']
    zero = datetime.strptime('01/01/2000', '%m/%d/%Y')

    def __init__(self, docs, window_size, delta):
        self.docs = docs
        self.delta = timedelta(days=delta)
        self.click_history = [0] * window_size
        self.click_history_time = [self.zero] * window_size

    def get_ids(self, time):
        return np.array([i if t >= time else 0 for i, t in zip(self.click_history, self.click_history_time)],
                        dtype=np.int32)

    def get_title(self, time):
        return self.docs.titles_f64[self.get_ids(time)]

    def push(self, doc, time):
        del self.click_history[0], self.click_history_time[0]
        self.click_history.append(doc)
        self.click_history_time.append(time + self.delta)

    @property
    def window_size(self):
        return len(self.click_history)

    def count(self, time):
        return sum(1 for t in self.click_history_time if t >= time)


class Seq2VecPaperSoftmaxDays(Seq2VecPaperSoftmax):
    """History entries expire `config.days` after the click (paper.py:668-750).  Same model as the parent."""
    Window = _ExpiringWindow

    def _sample_at(self, user, ch, pos, impression, label):
        cands = [pos] + [int(n) for n in impression.negative_samples(self.config.negative_samples)]
        head = (user,) if self.has_user else ()
        return head + (ch.get_ids(impression.time), cands, label)

    def _history_after_training(self, ih1):
        ch = self.Window(self.docs, self.config.window_size, self.config.days)
        for impression in ih1:
            for pos in impression.pos:
                ch.push(pos, impression.time)
        return ch

    def train_gen_ids(self):
        label = self._label()
        while True:
            emitted = False
            for user, (ih, _) in self._users(shard=True):
                if not ih:
                    continue
                ch = self.Window(self.docs, self.config.window_size, self.config.days)
                for impression in ih:
                    for pos in impression.pos:
                        if ch.count(impression.time):
                            emitted = True
                            yield self._sample_at(user, ch, pos, impression, label)
                        ch.push(pos, impression.time)
            if not emitted:
                raise self._no_samples()

    def valid_gen_ids(self):
        label = self._label()
        while True:
            for user, (ih1, ih2) in self._users():
                if not (ih1 and ih2):
                    continue
                ch = self._history_after_training(ih1)
                for impression in ih2:
                    for pos in impression.pos:
                        yield self._sample_at(user, ch, pos, impression, label)
                    for pos in impression.pos:
                        ch.push(pos, impression.time)

    def test_gen_ids(self):
        for user, (ih1, ih2) in self._users():
            if not (ih1 and ih2):
                continue
            ch = self._history_after_training(ih1)
            for impression in ih2:
                hist = ch.get_ids(impression.time)
                head = (user,) if self.has_user else ()
                yield ([head + (hist, [p], 1) for p in impression.pos] +
                       [head + (hist, [n], 0) for n in impression.neg])
                for pos in impression.pos:
                    ch.push(pos, impression.time)


class Seq2VecPaperSoftmaxDaysId(Seq2VecPaperSoftmaxDays):
    """`...Days` with the user-id model of Seq2VecPaperSoftmaxId (paper.py:753-881; the vertical / sub-vertical
    fields its News class also parses are not used by this task's generators or model)."""
    has_user = True
