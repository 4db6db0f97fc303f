"""Host utilities of the LSTUR path: embedding loader, ranking metrics, log lines.

Mirrors /root/reference/utils.py: ``load_textual_embedding`` (:34-52),
``dcg_score``/``ndcg_score``/``mrr_score`` (:106-124), ``logging_history`` (:17-23),
``logging_evaluation`` (:26-31).  ``open`` is the builtin (the reference aliases
tf.gfile.GFile, utils.py:14).
"""
import logging
import os

import numpy as np


def load_textual_embedding(path, dimension, native=False):
    """Vocab.tsv rows ``word \\t index \\t "v0 v1 ..."`` -> float32 [max_index+1, dimension].
    Rows whose vector does not have exactly ``dimension`` fields are skipped; missing
    indices (row 0 included) are zero.  native: the C++ loader (include/lstur_data.h), the same matrix."""
    logging.info('[+] loading embedding data from {}'.format(os.path.split(path)[-1]))
    if native:
        from .data_native import load_vocab
        matrix = load_vocab(path, dimension)
        logging.info('[-] loaded {} rows from {}'.format(len(matrix), os.path.split(path)[-1]))
        return matrix
    rows = {}
    with open(path) as f:
        for line in f:
            cols = line.strip().split('\t')
            if cols[-1].count(' ') != dimension - 1:
                continue
            rows[int(cols[-2])] = np.array([float(x) for x in cols[-1].split(' ')], dtype=np.float32)
    matrix = np.zeros((max(rows) + 1, dimension), dtype=np.float32)
    for idx, vec in rows.items():
        matrix[idx] = vec
    logging.info('[-] found {} vectors from {}'.format(len(rows), os.path.split(path)[-1]))
    return matrix


def dcg_score(y_true, y_score, k=10):
    top = np.argsort(y_score)[::-1][:k]
    gains = 2 ** np.take(y_true, top) - 1
    return np.sum(gains / np.log2(np.arange(len(top)) + 2))


def ndcg_score(y_true, y_score, k=10):
    return dcg_score(y_true, y_score, k) / dcg_score(y_true, y_true, k)


def mrr_score(y_true, y_score):
    ranked = np.take(y_true, np.argsort(y_score)[::-1])
    return np.sum(ranked / (np.arange(len(ranked)) + 1)) / np.sum(y_true)


def _log_line(pairs):
    logging.info('[*] {}'.format('\t'.join('{}: {:.4f}'.format(k, v) for k, v in pairs)))


def logging_history(history):
    """Last value per training key, mean per ``val*`` key (reference utils.py:17-23)."""
    try:
        _log_line((k, np.mean(v) if k.startswith('val') else v[-1]) for k, v in sorted(history.history.items()))
    except Exception:
        pass


def logging_evaluation(evaluations):
    try:
        _log_line(sorted(evaluations.items()))
    except Exception:
        pass
