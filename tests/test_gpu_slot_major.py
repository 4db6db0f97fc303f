"""The opt-in slot-major step (LSTUR_LATE_SLOTS=n / LSTUR_SLOT_MAJOR=1, include/lstur.h: lstur_slot_major): history titles
ordered (slot, sample) inside the workspace, the recurrence's backward cut at slot W - n with the late slots' encoder
backward (d(doc vectors) -> d(pooled) -> attention backward -> conv kernel gradient on title / token sub-ranges) forked
behind its first part.  It measures slower than the default schedule and stays off; these tests keep
it correct: parity with the oracle -- dropout on, so the masks drawn at the REFERENCE position of every element must
survive the re-ordering --, and agreement with the default layout after Adam steps."""
import numpy as np
import pytest

from conftest import make_batch
from test_gpu_model import MID, build, rel_close, _need_gpu

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.mark.parametrize('arch,dims,late', [('igru', MID, 3), ('gru', MID, 4), ('hgru', MID, 2), ('iigru', MID, 3), ('igru', MID, 0)])
def test_slot_major_step_matches_oracle_and_default_layout(arch, dims, late, monkeypatch):
    _need_gpu()
    from mnexp_amd._lib import lib, check
    B, W, T, K1, drop = 19, 9, 11, 5, 0.2
    outs = []
    for mode in ('default', 'slot'):
        if mode == 'slot':
            monkeypatch.setenv('LSTUR_LATE_SLOTS', str(late))
            monkeypatch.setenv('LSTUR_SLOT_MAJOR', '1')
        O, p, model, rng = build(arch, dims, B, W, T, K1, drop)
        assert lib.lstur_slot_major(model._ctx) == (1 if mode == 'slot' else 0)
        user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
        user[1] = user[0]
        hist[2] = 0                                    # a sample whose whole history is padding
        hist[3, :5] = 0                                # front-padded history: masked steps inside the late AND the early range
        u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
        model.step = 3
        ref = O.forward_backward(p, arch, user, hist, cands, drop_p=drop, seed=77, step=3)
        check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 3, 1.0,
                                         model.stats.data_ptr(), model._stream()), 'fb')
        torch.cuda.synchronize()
        np.testing.assert_allclose(model.peek('logits').cpu().numpy().reshape(B, K1), ref['logits'], rtol=1e-3, atol=1e-5)
        g = model.get_grads()
        for n in O.dense_param_names(arch):
            rel_close(g[n], ref['grads'][n], 2e-3, n)
        rel_close(g['user_row_grads'], ref['user_row_grads'], 2e-3, 'user_row_grads')
        for i in range(3):                             # three fused steps on top (Adam inside the call)
            model._train_step(u, h, c, False)
        torch.cuda.synchronize()
        outs.append({n: model.dense_param.cpu().numpy().copy() for n in ('dense',)} | {'user': model.user_emb.cpu().numpy().copy(),
                                                                                      'logits': model.peek('logits').cpu().numpy().copy()})
        del model
    for k in outs[0]:
        rel_close(outs[1][k], outs[0][k], 2e-3, k)
