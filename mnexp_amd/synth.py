"""Synthetic ClickData.tsv / DocMeta.tsv / Vocab.tsv writer.

Files follow the reference's on-disk formats (README.md:11-23 of the reference;
parsing sites task/seq2vec.py:85-110, task/paper.py:10-33, utils.py:34-52) with a
planted topic signal so a trained model reaches AUC well above 0.5.
"""
from __future__ import annotations

import os

import numpy as np

N_TOPICS = 32

PRESETS = {
    # config 1 of BASELINE.json (plumbing) -- also the fixture-capture set
    "tiny": dict(n_docs=200, vocab=500, dim=16, title_len=8, n_users=50, lam_train=6, lam_valid=2),
    # config 2/3
    "cfg2": dict(n_docs=20000, vocab=50000, dim=300, title_len=30, n_users=50000, lam_train=20, lam_valid=4),
    # config 4
    "cfg4": dict(n_docs=200000, vocab=100000, dim=300, title_len=30, n_users=1000000, lam_train=20, lam_valid=4),
}


def _zipf_probs(n, a=1.1):
    p = 1.0 / np.arange(1, n + 1) ** a
    return p / p.sum()


def write_vocab(path, vocab, dim, rng):
    with open(path, "w") as f:
        for s in range(1, vocab, 4096):
            e = min(vocab, s + 4096)
            vec = rng.normal(0.0, 0.1, size=(e - s, dim))
            for i in range(s, e):
                f.write("w%d\t%d\t%s\n" % (i, i, " ".join("%.6f" % x for x in vec[i - s])))


def make_docs(n_docs, vocab, title_len, rng):
    """Returns (topic[n_docs+1], list of title id lists); doc 0 is the reserved pad doc."""
    topic = rng.integers(0, N_TOPICS, size=n_docs + 1)
    slice_w = max(1, (vocab - 1) // N_TOPICS)
    zipf = _zipf_probs(vocab - 1)
    titles = [[]]
    lens = rng.integers(5, title_len + 6, size=n_docs + 1)
    for d in range(1, n_docs + 1):
        n = int(lens[d])
        from_topic = rng.random(n) < 0.7
        lo = 1 + int(topic[d]) * slice_w
        t_ids = rng.integers(lo, min(vocab, lo + slice_w), size=n)
        z_ids = 1 + rng.choice(vocab - 1, size=n, p=zipf)
        titles.append(np.where(from_topic, t_ids, z_ids).tolist())
    return topic, titles


def write_docmeta(path, titles):
    with open(path, "w") as f:
        for d in range(1, len(titles)):
            t = " ".join(map(str, titles[d]))
            f.write("id%d\t%d\tnews\tnewsus\t%s\t%s\n" % (d, d, t, t))


def _impressions(n_imp, mix, topic, doc_p, rng, stamp):
    out = []
    n_docs = len(topic) - 1
    for _ in range(n_imp):
        n = int(rng.integers(5, 21))
        shown = 1 + rng.choice(n_docs, size=n, p=doc_p)
        pclick = 1.0 / (1.0 + np.exp(-3.0 * (mix[topic[shown]] * N_TOPICS - 1.0)))
        click = rng.random(n) < pclick
        if not click.any():
            click[int(np.argmax(pclick))] = True
        if click.all():
            click[int(np.argmin(pclick))] = False
        pos = " ".join(map(str, shown[click]))
        neg = " ".join(map(str, shown[~click]))
        stamp[0] += int(rng.integers(60, 7200))
        s = stamp[0]
        day = 1 + (s // 86400) % 28
        hh = (s // 3600) % 24
        ampm = "AM" if hh < 12 else "PM"
        h12 = hh % 12 or 12
        ts = "01/%02d/2019 %02d:%02d:%02d %s" % (day, h12, (s // 60) % 60, s % 60, ampm)
        out.append("%s#TAB#%s#TAB#%s#TAB#x" % (pos, neg, ts))
    return "#N#".join(out)


def write_clickdata(path, n_users, topic, lam_train, lam_valid, rng):
    n_docs = len(topic) - 1
    doc_p = _zipf_probs(n_docs)
    doc_p = doc_p[rng.permutation(n_docs)]
    with open(path, "w") as f:
        for u in range(n_users):
            mix = rng.dirichlet(np.full(N_TOPICS, 0.3))
            stamp = [int(rng.integers(0, 86400))]
            n_tr = max(1, int(rng.poisson(lam_train)))
            n_va = 0 if rng.random() < 0.1 else max(1, int(rng.poisson(lam_valid)))
            tr = _impressions(n_tr, mix, topic, doc_p, rng, stamp)
            va = _impressions(n_va, mix, topic, doc_p, rng, stamp) if n_va else ""
            f.write("u%d\tt\t%s\t%s\n" % (u, tr, va))


def generate(out_dir, preset="tiny", seed=0, **over):
    """Write the three TSV files into out_dir; returns the resolved spec dict."""
    spec = dict(PRESETS[preset])
    spec.update(over)
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    write_vocab(os.path.join(out_dir, "Vocab.tsv"), spec["vocab"], spec["dim"], rng)
    topic, titles = make_docs(spec["n_docs"], spec["vocab"], spec["title_len"], rng)
    write_docmeta(os.path.join(out_dir, "DocMeta.tsv"), titles)
    write_clickdata(os.path.join(out_dir, "ClickData.tsv"), spec["n_users"], topic,
                    spec["lam_train"], spec["lam_valid"], rng)
    return spec
