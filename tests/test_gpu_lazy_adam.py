"""Lazy dense Adam of large user tables (csrc/misc.hip: lazy_*; include/lstur.h: lstur_flush_tables).  TF-1 Keras densifies
the embedding gradient (paper.py:657): every row takes an Adam update every step, a row outside the batch the zero-gradient
one.  The lazy path applies a row's pending zero-gradient updates when the row is next read or updated -- the same fp32
operations per element, so after a flush the table, its moments and everything trained on it must be BIT-identical to the
dense sweep.  Forced on a small table here (LSTUR_LAZY_ADAM=1 / 0 in child processes: the switch is read per context
through the environment of the process)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from test_gpu_model import MID, build
from mnexp_amd._lib import lib, check
arch, mode, out = sys.argv[1], sys.argv[2], sys.argv[3]
B, W, T, K1 = 12, 6, 9, 5
dims = dict(MID, N_u=97)
O, p, model, rng = build(arch, dims, B, W, T, K1, 0.2)
assert lib.lstur_lazy_adam(model._ctx) == int(os.environ['LSTUR_LAZY_ADAM'])
tt = np.random.default_rng(1).integers(1, dims['V'], size=(150, T)).astype(np.int32)
tt[0] = 0
model.set_title_table(tt)
r = np.random.default_rng(2)
losses, preds = [], []
for step in range(14):
    # users come from a sliding window of the table: some rows are hit in consecutive steps, most wait many steps, some
    # are never touched; duplicates inside a batch
    lo = (step * 7) %% 80
    user = r.integers(lo, lo + 17, size=B).astype(np.int32)
    user[3] = user[0]
    hist = r.integers(0, 150, size=(B, W)).astype(np.int32)
    cands = r.integers(1, 150, size=(B, K1)).astype(np.int32)
    u, h, c = (model._dev_i32(a) for a in (user, hist, cands))
    if step == 6:
        model.optimizer.lr = 4e-4                      # the per-epoch decay of the reference's callback (paper.py:498-499)
    if mode == 'fused':
        model._train_step(u, h, c, True)
    else:
        s = model._stream()
        check(lib.lstur_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, step, 1.0,
                                             model.stats.data_ptr(), s), 'fb')
        check(lib.lstur_adam_step(model._ctx, model.optimizer.lr, step + 1, None, None, 0, s), 'adam')
    losses.append(model.stats[0].item())
    if step in (4, 9):                                  # evaluation in between reads rows (and makes them current)
        preds.append(model.predict_ids(user, hist, cands))
model.flush_tables()
torch.cuda.synchronize()
names = ['dense_param', 'dense_m', 'dense_v', 'user_emb', 'user_m', 'user_v'] + (['user_emb2', 'user_m2', 'user_v2'] if arch == 'iigru' else [])
np.savez(out, loss=np.array(losses), pred=np.concatenate(preds), **{n: getattr(model, n).cpu().numpy() for n in names})
'''


@pytest.mark.parametrize('arch,mode', [('igru', 'fused'), ('igru', 'split'), ('iigru', 'fused'), ('gru', 'split')])
def test_lazy_adam_is_bit_identical_to_the_dense_sweep(arch, mode, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    code = _CHILD % (ROOT, os.path.join(ROOT, 'tests'))
    outs = []
    for lazy in ('1', '0'):
        out = str(tmp_path / ('lazy%s.npz' % lazy))
        env = dict(os.environ, LSTUR_LAZY_ADAM=lazy)
        r = subprocess.run([sys.executable, '-c', code, arch, mode, out], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        outs.append(np.load(out))
    for k in outs[0].files:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert len(set(outs[0]['loss'])) > 5
