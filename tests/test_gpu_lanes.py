"""The two-lane training step (csrc/lstur_capi.hip: train_two_lanes): a batch of >= 32 samples is cut into two sample
ranges that run one stage apart on two streams, so that each lane's GRU recurrence runs beside the other lane's encoder.
Parity against the oracle's restatement of the same split (oracle.forward_backward_lanes: identical mathematics, lane 1
draws its dropout masks from its own streams), bit-identity of the two-stream run with the one-stream run of the same
lanes, and agreement with the uncut step.  Reference graph: task/paper.py:635-665."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import make_batch
from test_gpu_model import MID, build, rel_close, _need_gpu

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _step(model, user, hist, cands, B):
    from mnexp_amd._lib import lib, check
    u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 0, 1.0,
                                     model.stats.data_ptr(), model._stream()))
    torch.cuda.synchronize()


@pytest.mark.parametrize('arch,B,drop', [('igru', 64, 0.2), ('igru', 40, 0.0), ('gru', 48, 0.2), ('hgru', 33, 0.2),
                                         ('nigru', 64, 0.2), ('dgru', 32, 0.2), ('iigru', 37, 0.2)])
def test_two_lane_step_matches_oracle(arch, B, drop, monkeypatch):
    _need_gpu()
    monkeypatch.setenv('LSTUR_PIPELINE', '1')
    W, T, K1 = 7, 11, 5
    from oracle import lstur_oracle as O
    from mnexp_amd.native_model import NativeModel
    sm = 'ddot' if arch == 'dgru' else 'dot'               # dgru's user vector is 2U wide (paper.py:608-611)
    rng = np.random.default_rng(5)
    p = O.init_params(rng, arch=arch, dtype=np.float64, score_model=sm, **MID)
    for n in p:
        if n.endswith('_b'):
            p[n] = rng.normal(0, 0.1, size=p[n].shape)
    model = NativeModel(p['word_emb'], MID['N_u'], B, W, T, K1 - 1, MID['F'], MID['k'], MID['U'], arch=arch, dropout=drop,
                        lr=1e-3, seed=77, score_model=sm)
    model.set_weights({n: p[n] for n in p if n != 'word_emb'})
    assert O.lane_split(B) < B
    user, hist, cands = make_batch(rng, B, W, T, K1, MID['V'], MID['N_u'])
    user[3] = user[B - 2]                                  # the same user in both lanes
    ref = O.forward_backward_lanes(p, arch, user, hist, cands, drop_p=drop, seed=77, step=0)
    _step(model, user, hist, cands, B)
    logits = model.peek('logits').cpu().numpy().reshape(B, K1)
    np.testing.assert_allclose(logits, ref['logits'], rtol=1e-3, atol=1e-5)
    assert abs(model.stats[0].item() - ref['loss']) < 1e-4
    g = model.get_grads()
    for n in O.dense_param_names(arch, score_model=sm):
        rel_close(g[n], ref['grads'][n], 2e-3, n)
    if arch != 'nigru':
        assert np.array_equal(g['user_rows'], user)
        rel_close(g['user_row_grads'], ref['user_row_grads'], 2e-3, 'user_row_grads')
    if arch == 'iigru':
        rel_close(model.peek('duser2').cpu().numpy().reshape(B, -1), ref['user_row_grads2'], 2e-3, 'user_row_grads2')
    with pytest.raises(Exception, match='two-lane'):
        model.peek('doc_vec')


def test_two_lane_step_without_dropout_is_the_uncut_step(monkeypatch):
    """With dropout off nothing depends on the lane layout: the two-lane step agrees with the UNCUT oracle step (fp32
    summation-order noise only)."""
    _need_gpu()
    monkeypatch.setenv('LSTUR_PIPELINE', '1')
    B, W, T, K1 = 64, 7, 11, 5
    O, p, model, rng = build('igru', MID, B, W, T, K1, 0.0)
    user, hist, cands = make_batch(rng, B, W, T, K1, MID['V'], MID['N_u'])
    ref = O.forward_backward(p, 'igru', user, hist, cands, drop_p=0.0, seed=77, step=0)
    _step(model, user, hist, cands, B)
    g = model.get_grads()
    for n in O.dense_param_names('igru'):
        rel_close(g[n], ref['grads'][n], 2e-3, n)
    rel_close(g['user_row_grads'], ref['user_row_grads'], 2e-3, 'user_row_grads')


_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import make_batch
from test_gpu_model import MID, build
from mnexp_amd._lib import lib, check
B, W, T, K1 = 64, 7, 11, 5
O, p, model, rng = build('igru', MID, B, W, T, K1, 0.2)
user, hist, cands = make_batch(rng, B, W, T, K1, MID['V'], MID['N_u'])
u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
for step in range(3):
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, step, 1.0,
                                     model.stats.data_ptr(), model._stream()))
    check(lib.lstur_adam_step(model._ctx, 1e-3, step + 1, None, None, 0, model._stream()))
torch.cuda.synchronize()
assert model.sizes.workspace_bytes > 0
try:
    model.peek('doc_vec')
    raise SystemExit('the step did not take two lanes')
except Exception as e:
    assert 'two-lane' in str(e), e
np.savez(sys.argv[1], grad=model.dense_grad.cpu().numpy(), param=model.dense_param.cpu().numpy(),
         user=model.user_emb.cpu().numpy(), logits=model.peek('logits').cpu().numpy())
'''


def test_two_streams_are_bit_identical_to_one_stream(tmp_path):
    """The same two lanes on ONE stream (LSTUR_STREAMS=1: same kernels, same data, no overlap) give bit-identical
    gradients, parameters after three Adam steps, user table and logits: the overlap changes timing, never a result."""
    _need_gpu()
    outs = []
    for tag, env in (('two', {'LSTUR_PIPELINE': '1'}), ('one', {'LSTUR_PIPELINE': '1', 'LSTUR_STREAMS': '1'})):
        f = str(tmp_path / (tag + '.npz'))
        e = dict(os.environ, **env)
        subprocess.check_call([sys.executable, '-c', _CHILD % (ROOT, os.path.join(ROOT, 'tests')), f], env=e, cwd=ROOT)
        outs.append(np.load(f))
    for k in ('grad', 'param', 'user', 'logits'):
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_prefetched_staging_is_bit_identical_to_inline_staging():
    """lstur_prefetch_ids: the next batch's title expansion, token compaction and embedding gather (dropout included) run on
    a side stream under the current step's recurrence.  Six steps with every batch announced one step ahead (one of them
    announced and then NOT used: the library falls back to inline staging) against six steps staged inline: identical
    gradients, parameters, user table and losses."""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    B, W, T, K1 = 24, 7, 11, 5
    outs = []
    for mode in ('prefetch', 'inline'):
        O, p, model, rng = build('igru', MID, B, W, T, K1, 0.2)
        tt = np.random.default_rng(1).integers(1, MID['V'], size=(200, T)).astype(np.int32)
        tt[0] = 0
        tt[:, 8:] = 0
        model.set_title_table(tt)
        r2 = np.random.default_rng(2)
        batches = [(r2.integers(0, MID['N_u'], size=B).astype(np.int32), r2.integers(0, 200, size=(B, W)).astype(np.int32),
                    r2.integers(1, 200, size=(B, K1)).astype(np.int32)) for _ in range(6)]
        dev = [tuple(model._dev_i32(a) for a in b) for b in batches]
        decoy = tuple(model._dev_i32(a) for a in batches[0])
        losses = []
        for i, (u, h, c) in enumerate(dev):
            if mode == 'prefetch' and i + 1 < len(dev):
                nh, nc = (decoy[1], decoy[2]) if i == 2 else (dev[i + 1][1], dev[i + 1][2])    # step 3 was mis-announced
                model.prefetch_ids(nh, nc)
            model._train_step(u, h, c, True)
            losses.append(model.stats[0].item())
        torch.cuda.synchronize()
        # steps 2, 3, 5, 6 ran on a prefetched batch; step 4 was mis-announced (staged inline, the staged decoy dropped)
        assert lib.lstur_prefetch_hits(model._ctx) == (4 if mode == 'prefetch' else 0)
        outs.append(dict(loss=np.array(losses), grad=model.dense_grad.cpu().numpy(), param=model.dense_param.cpu().numpy(),
                         user=model.user_emb.cpu().numpy()))
        del model
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert np.isfinite(outs[0]['loss']).all() and len(set(outs[0]['loss'])) > 1
