"""Capture golden vectors from the REFERENCE's pure-Python data path (this container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference; never runs on the GPU box)

The reference's numeric path needs keras/tensorflow, which are absent; its data path
(parsing, windowing, negative sampling, pool batching, ranking metrics) is plain
Python/NumPy and runs once ``keras``/``tensorflow`` are present as *names*.  The
stand-ins below carry no behaviour beyond ``open``/``makedirs``/``listdir`` aliases
(SURVEY.md Appendix B).  Outputs are DATA only (inputs + expected outputs):
  tests/golden/tiny/{ClickData,DocMeta,Vocab}.tsv   synthetic inputs (mnexp_amd.synth, seed 0)
  tests/golden/edge/...                             hand-written edge-case inputs
  tests/golden/data_path.npz                        what the reference produced on them
  tests/golden/metrics.json                         dcg/ndcg/mrr known answers
"""
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def install_stubs():
    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Any()

        def __getattr__(self, n):
            return _Any()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__getattr__ = lambda n: _Any()
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    layers = mod('keras.layers', Layer=type('Layer', (), {'__init__': lambda self, **k: None}))
    backend = mod('keras.backend')
    mod('keras', layers=layers, backend=backend)
    gfile = types.SimpleNamespace(GFile=open, MkDir=lambda p: os.makedirs(p, exist_ok=True),
                                  ListDirectory=os.listdir, Exists=os.path.exists)
    errors = types.SimpleNamespace(NotFoundError=FileNotFoundError)
    mod('tensorflow', gfile=gfile, errors=errors)


EDGE_DOCS = [
    # id, idx, vert, subvert, title, body
    ("a", 1, "1 2 3", "1"),
    ("b", 2, "4 5#N#6 7 8", "2"),                    # only the first '#N#' segment is kept (an empty one makes the reference raise)
    ("c", 3, "9 10 11 12 13 14 15 16 17 18", "3"),   # longer than T -> truncated
    ("d", 5, "20", "4"),                             # hole at idx 4
    ("e", 7, "21 22#N#23", "5"),
    ("f", 7, "31 32 33", "6"),                       # duplicate idx: later row wins
]
EDGE_CLICKS = [
    "u0\tt\t1 2#TAB#3 5#TAB#01/18/2019 08:15:31 am#TAB#x#N#3#TAB#1 7 5#TAB#01/18/2019 08:54:18 am\t5#TAB#1 2 3#TAB#01/19/2019 01:00:00 pm",
    "u1\tt\t\t",                                                      # empty train + valid
    "u2\tt\t#TAB#1 2#TAB#01/18/2019 08:15:31 am#N#7 5 3#TAB#2#TAB#01/18/2019 09:15:31 am\t",   # first impression has no positives -> dropped; no valid
    "u3\tt\t1#TAB#2\t#TAB#3#N#2 3#TAB#5 7",                           # no time field; valid: dropped + kept
    "u4\tt\t1 2 3 5 7 1 2 3#TAB#5\t7#TAB#1 2#N#1#TAB#7",              # history longer than window
]


def write_edge(d):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, 'DocMeta.tsv'), 'w') as f:
        for name, idx, title, body in EDGE_DOCS:
            f.write('%s\t%d\tnews\tnewsus\t%s\t%s\n' % (name, idx, title, body))
    with open(os.path.join(d, 'ClickData.tsv'), 'w') as f:
        for line in EDGE_CLICKS:
            f.write(line + '\n')
    with open(os.path.join(d, 'Vocab.tsv'), 'w') as f:
        f.write('w1\t1\t0.5 -0.25 1e-3 2\n')
        f.write('w3\t3\t1 2 3\n')                      # wrong dimension -> skipped
        f.write('w4\t4\t-1.5 0.1 0.2 0.3\n')
        f.write('w6\t6\t0.333333 0.1234567 7.7e-5 -0\n')


def ref_config(settings, data_dir, **kw):
    settings.public_config = dict(input_training_data_path=data_dir, input_validation_data_path=data_dir,
                                  input_previous_model_path='/tmp/_gold_models', output_model_path='/tmp/_gold_models',
                                  log_dir='/tmp/_gold_logs')
    base = dict(task='Seq2VecPaperSoftmaxId', arch='igru', batch_size=8, window_size=5, title_shape=8,
                body_shape=4, negative_samples=4, training_step=20, validation_step=2, max_impression=200,
                news_encoder='cnnatt', textual_embedding_dim=16, days=30, debug=False, epochs=2,
                validation_impression=10, testing_impression=10, learning_rate=1e-3, learning_rate_decay=0.2,
                dropout=0.0, user_embedding_dim=8, title_filter_shape=(16, 3), score_model='dot',
                enable_pretrain_encoder=False, textual_embedding_trainable=False, name='')
    base.update(kw)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return settings.Config(base)


def capture(handler, n_samples, n_batches, seed, has_user=True):
    out = {}
    u = 1 if has_user else 0          # id-less tasks carry no user column
    np.random.seed(seed)
    random.seed(seed)
    gen = handler.train_gen()
    samples = [next(gen) for _ in range(n_samples)]
    if has_user:
        out['train_user'] = np.array([s[0] for s in samples], dtype=np.int64)
    out['train_clicked'] = np.stack([s[u] for s in samples])
    out['train_cands'] = np.stack([np.stack(s[u + 1:-1]) for s in samples])
    out['train_label'] = np.array([s[-1] for s in samples])

    np.random.seed(seed + 1)
    random.seed(seed + 1)
    it = handler.train
    for b in range(n_batches):
        X, Y = next(it)
        if has_user:
            out['batch%d_user' % b] = X[0]
        out['batch%d_clicked' % b] = X[u]
        out['batch%d_cands' % b] = np.stack(X[u + 1:], axis=1)
        out['batch%d_y' % b] = Y
    out['rng_after_batches'] = np.random.randint(0, 1 << 30, size=4)

    np.random.seed(seed + 2)
    it = handler.valid
    for b in range(3):
        X, Y = next(it)
        if has_user:
            out['valid%d_user' % b] = X[0]
        out['valid%d_clicked' % b] = X[u]
        out['valid%d_cands' % b] = np.stack(X[u + 1:], axis=1)
        out['valid%d_y' % b] = Y

    tests = []
    for i, rows in zip(range(12), handler.test_gen()):
        cols = [np.stack(x) for x in zip(*rows)]
        tests.append(cols)
        if has_user:
            out['test%d_user' % i] = cols[0]
        out['test%d_clicked' % i] = cols[u]
        out['test%d_cand' % i] = cols[u + 1]
        out['test%d_label' % i] = cols[u + 2]
    out['n_test'] = np.array(len(tests))
    return out


def main():
    if not os.path.isdir(REF):
        raise SystemExit('reference tree not present: golden vectors can only be captured in the build container')
    install_stubs()
    sys.path.insert(0, REF)
    import document as ref_document          # noqa  (reference modules)
    import settings as ref_settings
    import task as ref_task
    import utils as ref_utils
    from mnexp_amd import synth

    tiny = os.path.join(HERE, 'tiny')
    edge = os.path.join(HERE, 'edge')
    synth.generate(tiny, 'tiny', seed=0)
    write_edge(edge)

    out = {}
    # --- tiny set -----------------------------------------------------------
    cfg = ref_config(ref_settings, tiny)
    h = ref_task.get(cfg)
    n_docs = h.doc_count
    out['tiny_titles'] = np.stack([h.docs[i].title if i in h.docs else np.full(8, -1.0) for i in range(n_docs)])
    out['tiny_present'] = np.array([i in h.docs for i in range(n_docs)])
    out['tiny_embedding'] = ref_utils.load_textual_embedding(cfg.title_embedding_input, 16)
    out['tiny_n_users'] = np.array(len(h.data))
    for k, v in capture(h, 200, 10, seed=11).items():
        out['tiny_' + k] = v
    # --- edge set -----------------------------------------------------------
    cfg = ref_config(ref_settings, edge, window_size=3, title_shape=6, batch_size=2, negative_samples=2,
                     textual_embedding_dim=4)
    h = ref_task.get(cfg)
    out['edge_titles'] = np.stack([h.docs[i].title if i in h.docs else np.full(6, -1.0) for i in range(h.doc_count)])
    out['edge_present'] = np.array([i in h.docs for i in range(h.doc_count)])
    out['edge_embedding'] = ref_utils.load_textual_embedding(cfg.title_embedding_input, 4)
    out['edge_imp_counts'] = np.array([[len(a), len(b)] for a, b in h.data])
    for k, v in capture(h, 30, 4, seed=5).items():
        out['edge_' + k] = v
    # --- time-window variants on the tiny set (days = 0.05 so entries do expire inside a user's history)
    for tname in ('Seq2VecPaperSoftmaxDaysId', 'Seq2VecPaperSoftmaxDays'):
        cfg = ref_config(ref_settings, tiny, task=tname, days=0.05)
        h = ref_task.get(cfg)
        for k, v in capture(h, 120, 4, seed=21, has_user=tname.endswith('Id')).items():
            out['%s_%s' % (tname, k)] = v
    # --- pointwise sibling on the tiny set: max_impression = 3 so random.sample() picks subsets
    for tname in ('Seq2VecPaperId', 'Seq2VecPaper'):
        cfg = ref_config(ref_settings, tiny, task=tname, arch='igru' if tname.endswith('Id') else 'gru',
                         max_impression=3)
        h = ref_task.get(cfg)
        for k, v in capture(h, 150, 4, seed=31, has_user=tname.endswith('Id')).items():
            out['%s_%s' % (tname, k)] = v
    np.savez_compressed(os.path.join(HERE, 'data_path.npz'), **out)

    # --- parser known answers -------------------------------------------------
    pd = ref_document.DocumentParser(ref_document.parse_document(), ref_document.pad_document(2, 4))
    parsed = {s: pd(s).tolist() for s in ['1 2 3', '4 5#N#6 7 8 9 10', '7', '1 2#N#3 4#N#5 6']}

    # --- ranking metrics ------------------------------------------------------
    rng = np.random.RandomState(3)
    cases = []
    for n in (2, 5, 9, 20):
        for _ in range(3):
            y = (rng.rand(n) < 0.4).astype(int)
            y[rng.randint(n)] = 1
            s = np.round(rng.rand(n), 2)          # rounded -> ties
            cases.append(dict(y_true=y.tolist(), y_score=s.tolist(),
                              dcg10=float(ref_utils.dcg_score(y, s, 10)),
                              ndcg10=float(ref_utils.ndcg_score(y, s, 10)),
                              ndcg5=float(ref_utils.ndcg_score(y, s, 5)),
                              mrr=float(ref_utils.mrr_score(y, s))))
    with open(os.path.join(HERE, 'metrics.json'), 'w') as f:
        json.dump(dict(parsed=parsed, ranking=cases), f, indent=1)
    print('golden fixtures written:', sorted(os.listdir(HERE)))


if __name__ == '__main__':
    main()
