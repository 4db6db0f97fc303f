"""Seq2Vec base task: document table, click window, batchers.

Host-side mirror of the parts of /root/reference/task/seq2vec.py the LSTUR path
inherits: ``Window`` (:17-53), ``Impression`` (:55-64), ``News`` (:66-71),
``_load_docs`` (:85-110), the shuffle-pool batcher ``train`` (:182-193), ``valid``
(:195-200), ``test`` (:202-206), ``build_model`` (:208-211).

Design difference (MI355X-first): samples are carried in *doc-id form*
(user, int32[W] history ids, K+1 candidate ids) and titles live in ONE int32 matrix
``title_table[doc_count, T]`` that is also resident in HBM; the reference's expanded
float64 title tensors are produced only at the ``train``/``valid``/``test`` boundary
(``titles_f64[ids]``), bit-identical to the reference's ``np.stack`` of per-doc rows.
The NumPy global RNG is consumed in exactly the reference's order (one
``np.random.choice`` per sample, one ``np.random.shuffle`` of the pool per batch).
"""
import logging

import numpy as np

from .. import document, utils


class DocTable:
    """idx -> News view over one [doc_count, T] matrix (doc 0 = all-zero pad title)."""

    def __init__(self, titles_i32, present):
        self.title_table = titles_i32                     # int32 [doc_count, T]
        self.titles_f64 = titles_i32.astype(np.float64)   # reference dtype (document.py:39)
        self.present = present

    def __getitem__(self, idx):
        if idx < 0 or idx >= len(self.present) or not self.present[idx]:
            raise KeyError(idx)
        return Seq2Vec.News(self.titles_f64[idx], None)

    def __len__(self):
        return int(self.present.sum())

    def keys(self):
        return np.flatnonzero(self.present).tolist()

    def check(self, ids):
        ids = np.asarray(ids)
        if ids.size and (ids.min() < 0 or ids.max() >= len(self.present) or not self.present[ids].all()):
            bad = ids[(ids < 0) | (ids >= len(self.present))]
            raise KeyError(int(bad[0]) if bad.size else int(ids[~self.present[ids]][0]))


class Seq2Vec:
    class Window:
        """Last ``window_size`` clicked doc ids, front-padded with 0, newest last."""
        __slots__ = ['count', 'docs', 'click_history']

        def __init__(self, docs, window_size):
            self.count = 0
            self.docs = docs
            self.click_history = [0] * window_size

        def get_ids(self):
            return np.array(self.click_history, dtype=np.int32)

        def get_title(self):
            return self.docs.titles_f64[self.click_history]

        def push(self, doc):
            del self.click_history[0]
            self.click_history.append(doc)
            self.count += 1

        @property
        def window_size(self):
            return len(self.click_history)

    class Impression:
        __slots__ = ['pos', 'neg']

        def __init__(self, d):
            fields = d.split('#TAB#')
            self.pos = [int(k) for k in fields[0].split(' ')]
            self.neg = [int(k) for k in fields[1].split(' ')]

        def negative_samples(self, n):
            return np.random.choice(self.neg, n)      # with replacement, global legacy RNG

    class News:
        __slots__ = ['title', 'body']

        def __init__(self, title, body):
            self.title = title
            self.body = body

    def __init__(self, config):
        self.is_training = True
        self.config = config
        self.model = None
        self._load_docs()
        self._load_users()
        self._load_data()

    # ---- loading -----------------------------------------------------------
    def _load_docs(self):
        """DocMeta.tsv: col[1] doc index, col[4] title ids; first non-empty '#N#' segment kept."""
        logging.info("[+] loading docs metadata")
        T = self.config.title_shape
        if getattr(self.config, 'native_load', False):     # the same table from the C++ loader (include/lstur_data.h)
            from ..data_native import load_docs
            table, present = load_docs(self.config.doc_meta_input, T)
            self.doc_count = len(present)
            self.docs = DocTable(table, present)
            logging.info("[-] loaded docs metadata")
            return
        title_parser = document.DocumentParser(document.parse_document(), document.pad_document(1, T))
        rows = {}
        with open(self.config.doc_meta_input) as file:
            for line in file:
                cols = line.strip('\n').split('\t')
                rows[int(cols[1])] = title_parser(cols[4])[0]
        self.doc_count = max(rows) + 1
        table = np.zeros((self.doc_count, T), dtype=np.int32)
        present = np.zeros(self.doc_count, dtype=bool)
        for idx, row in rows.items():
            table[idx] = row
            present[idx] = True
        table[0] = 0                                   # reference overwrites doc 0 with zeros (:106-108)
        present[0] = True
        self.docs = DocTable(table, present)
        logging.info("[-] loaded docs metadata")

    def _load_users(self):
        pass

    def _load_data(self):
        self.training_step = self.config.training_step
        self.validation_step = self.config.validation_step

    # ---- id-form -> reference-form ----------------------------------------
    def _expand(self, batch_ids):
        """[(user?, hist_ids, cand_ids..., label)] columns -> reference X list / Y."""
        raise NotImplementedError

    def _pool_batches(self, gen):
        """Shuffle pool of >= 100*B samples; emit the first B after every shuffle (seq2vec.py:182-193)."""
        pool = []
        B = self.config.batch_size
        size = B * 100
        while True:
            pool.append(next(gen))
            if len(pool) >= size:
                np.random.shuffle(pool)
                yield pool[:B]
                pool = pool[B:]

    @property
    def train(self):
        for samples in self._pool_batches(self.train_gen_ids()):
            yield self._expand(samples)

    @property
    def train_ids(self):
        for samples in self._pool_batches(self.train_gen_ids()):
            yield self._stack_ids(samples)

    @property
    def valid(self):
        gen = self.valid_gen_ids()
        B = self.config.batch_size
        while True:
            yield self._expand([next(gen) for _ in range(B)])

    @property
    def valid_ids(self):
        feed = self.native_feed()
        if feed is not None:                   # the same batches from the C++ feed (include/lstur_data.h)
            yield from feed.valid_batches()
        gen = self.valid_gen_ids()
        B = self.config.batch_size
        while True:
            yield self._stack_ids([next(gen) for _ in range(B)])

    def native_feed(self):
        """the task's C++ sample feed (None: the Python generators) -- tasks with a natively loaded ClickData override"""
        return None

    def test_items(self, limit=None, rank=0, world=1):
        """(i, scores, labels) of validation impression i for i = rank, rank + world, ... < limit: `test` with the
        impressions dealt out over the ranks of a data-parallel run (world == 1: every impression)."""
        native = hasattr(self.model, 'predict_ids')
        feed = self.native_feed()
        if feed is not None:
            it = feed.test_impressions(rank, world) if world > 1 else enumerate(feed.test_impressions())
            for i, (ids, labels) in it:
                if limit is not None and i >= limit:
                    return
                yield i, self.model.predict_ids(*ids).reshape(-1), labels
            return
        for i, rows in enumerate(self.test_gen_ids()):
            if limit is not None and i >= limit:
                return
            if i % world != rank:
                continue
            if native:
                ids, labels = self._stack_ids(rows)
                yield i, self.model.predict_ids(*ids).reshape(-1), labels
            else:
                X, labels = self._expand(rows)
                yield i, self.model.predict(X).reshape(-1), labels

    @property
    def test(self):
        """Per validation impression: [scores, labels] (seq2vec.py:202-206)."""
        for _, y_pred, y_true in self.test_items():
            yield [y_pred, y_true]

    def build_model(self, epoch):
        if epoch == 0:
            self._build_model()
        return self.model

    def save_model(self):
        logging.info('[+] saving models')
        utils_save = getattr(self.model, 'save', None)
        if utils_save is not None:
            utils_save(self.config.model_output)
        logging.info('[-] saved models')
