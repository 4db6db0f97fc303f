"""Task registry: ``task.get(config)`` -> task handler (reference task/__init__.py:15-16,
with the ``eval`` of the class name replaced by a lookup table)."""
from .paper import (Seq2VecPaper, Seq2VecPaperDot, Seq2VecPaperId, Seq2VecPaperSoftmax, Seq2VecPaperSoftmaxDays,
                    Seq2VecPaperSoftmaxDaysId, Seq2VecPaperSoftmaxId)
from .seq2vec import Seq2Vec

REGISTRY = {cls.__name__: cls for cls in (Seq2VecPaper, Seq2VecPaperDot, Seq2VecPaperId, Seq2VecPaperSoftmax,
                                          Seq2VecPaperSoftmaxId, Seq2VecPaperSoftmaxDays,
                                          Seq2VecPaperSoftmaxDaysId)}


def get(config):
    try:
        cls = REGISTRY[config.task]
    except KeyError:
        raise KeyError('task {!r} is not part of the native LSTUR path; available: {}'.format(
            config.task, sorted(REGISTRY)))
    return cls(config)
