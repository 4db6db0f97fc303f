"""Token-string -> padded id-matrix parsers.

Host-side mirror of the two reference parsers the LSTUR path uses
(/root/reference/document.py:23-27 ``parse_document`` and :37-54 ``pad_document``,
chained by ``DocumentParser`` :12-20).  Output dtype is float64 like the
reference (``np.zeros`` default), values are integer ids.
"""
import numpy as np


class DocumentParser:
    """Left-to-right composition of parser stages."""

    def __init__(self, *stages):
        self.stages = stages

    def __call__(self, doc):
        for stage in self.stages:
            doc = stage(doc)
        return doc


def parse_document(sep1='#N#', sep2=' '):
    """"12 7#N#99" -> [[12, 7], [99]]  (int() raises on an empty segment, as in the reference)."""
    def parse(text):
        return [[int(tok) for tok in seg.split(sep2)] for seg in text.split(sep1)]

    return parse


def pad_document(size, length):
    """Keep the first ``size`` non-empty segments, each truncated / zero-padded to ``length``."""
    def pad(segments):
        out = np.zeros((size, length))
        row = 0
        for seg in segments:
            if not seg:
                continue
            n = min(len(seg), length)
            out[row, :n] = seg[:n]
            row += 1
            if row == size:
                break
        return out

    return pad
