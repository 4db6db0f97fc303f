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
arch, mode, out, nsteps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
B, W, T, K1 = 12, 6, 9, 5
dims = dict(MID, N_u=97)
O, p, model, rng = build(arch, dims, B, W, T, K1, 0.2)
assert lib.lstur_lazy_adam(model._ctx) == int(os.environ['LSTUR_LAZY_ADAM'])
tt = np.random.default_rng(1).integers(1, dims['V'], size=(150, T)).astype(np.int32)
tt[0] = 0
model.set_title_table(tt)
r = np.random.default_rng(2)
batches = []
for step in range(nsteps + 2):
    # users come from a sliding window of the table: some rows are hit in consecutive steps (a row of batch i AND i + 1 must
    # take its catch-up and its step in order), most wait many steps -- longer than a short lr_t ring --, some are never
    # touched; duplicates inside a batch
    lo = (step * 7) %% 80
    user = r.integers(lo, lo + 17, size=B).astype(np.int32)
    user[3] = user[0]
    if batches:
        user[1] = batches[-1][0][0]
    hist = r.integers(0, 150, size=(B, W)).astype(np.int32)
    cands = r.integers(1, 150, size=(B, K1)).astype(np.int32)
    batches.append((user, hist, cands))
dev = [tuple(model._dev_i32(a) for a in b) for b in batches]
losses, preds = [], []
for step in range(nsteps):
    user, hist, cands = batches[step]
    u, h, c = dev[step]
    if step == 6:
        model.optimizer.lr = 4e-4                      # the per-epoch decay of the reference's callback (paper.py:498-499)
    if mode == 'prefetch':
        # the next batch announced WITH its users (lstur_prefetch_rows: their rows catch up under this step's recurrence, on
        # the parameter-gradient stream); step 5 announces the wrong batch (it is then staged inline), step 9 none
        nxt = dev[step + 2] if step == 5 else dev[step + 1]
        if step != 9:
            model.prefetch_ids(nxt[1], nxt[2], nxt[0])
    if mode in ('fused', 'prefetch'):
        model._train_step(u, h, c, True)
    else:
        s = model._stream()
        check(lib.lstur_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, step, 1.0,
                                             model.stats.data_ptr(), s), 'fb')
        check(lib.lstur_adam_step(model._ctx, model.optimizer.lr, step + 1, None, None, 0, s), 'adam')
    losses.append(model.stats[0].item())
    if step in (4, 9, 17):                              # evaluation in between reads rows (and makes them current)
        preds.append(model.predict_ids(user, hist, cands))
if mode == 'prefetch':
    assert lib.lstur_prefetch_hits(model._ctx) >= nsteps - 4, lib.lstur_prefetch_hits(model._ctx)
model.flush_tables()
torch.cuda.synchronize()
names = ['dense_param', 'dense_m', 'dense_v', 'user_emb', 'user_m', 'user_v'] + (['user_emb2', 'user_m2', 'user_v2'] if arch == 'iigru' else [])
np.savez(out, loss=np.array(losses), pred=np.concatenate(preds), **{n: getattr(model, n).cpu().numpy() for n in names})
'''


@pytest.mark.parametrize('arch,mode,ring,nsteps', [
    ('igru', 'fused', 0, 14), ('igru', 'split', 0, 14), ('iigru', 'fused', 0, 14), ('gru', 'split', 0, 14),
    # the prefetch-time catch-up of the NEXT batch's rows (lstur_prefetch_rows) on the parameter-gradient stream,
    # users overlapping between consecutive batches, one and two tables, a mis-announced batch
    ('igru', 'prefetch', 0, 14), ('iigru', 'prefetch', 0, 14),
    # the lr_t ring WRAPS -- a ring of 8 entries (LSTUR_LAZY_LOG=8; the product ring holds 65 536) forces a full
    # flush every 6 iterations: 30 steps = 5 flushes, rows that wait longer than the ring, fused / split / prefetched steps
    ('igru', 'fused', 8, 30), ('iigru', 'split', 8, 30), ('igru', 'prefetch', 8, 30), ('gru', 'fused', 16, 40),
])
def test_lazy_adam_is_bit_identical_to_the_dense_sweep(arch, mode, ring, nsteps, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    code = _CHILD % (ROOT, os.path.join(ROOT, 'tests'))
    outs = []
    for lazy in ('1', '0'):
        out = str(tmp_path / ('lazy%s.npz' % lazy))
        env = dict(os.environ, LSTUR_LAZY_ADAM=lazy)
        if ring and lazy == '1':
            env['LSTUR_LAZY_LOG'] = str(ring)
        r = subprocess.run([sys.executable, '-c', code, arch, mode, out, str(nsteps)], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        outs.append(np.load(out))
    for k in outs[0].files:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert len(set(outs[0]['loss'])) > 5
