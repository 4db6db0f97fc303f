"""lstur_train_step(_ids): forward + backward + Adam of one process in ONE library call (include/lstur.h).  The library
schedules the optimizer itself -- the dense sweep of the user table(s) on a side stream under the encoder backward, the
next step's weight images packed behind the dense Adam -- and must produce exactly what the split calls
(lstur_forward_backward + lstur_adam_step) produce.  Reference: model.train_on_batch inside fit_generator, main.py:73-78."""
import numpy as np
import pytest

from test_gpu_model import MID, ODD, build, _need_gpu

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _batches(model, B, W, K1, T, n_steps, n_users, n_docs=200, vocab=MID['V']):
    tt = np.random.default_rng(1).integers(1, vocab, size=(n_docs, T)).astype(np.int32)
    tt[0] = 0
    tt[:, 8:] = 0
    model.set_title_table(tt)
    r = np.random.default_rng(2)
    out = []
    for _ in range(n_steps):
        u = r.integers(0, n_users, size=B).astype(np.int32)
        u[5] = u[1]                        # duplicate user rows: combined in sample order by both paths
        u[B - 1] = u[1]
        out.append(tuple(model._dev_i32(a) for a in (u, r.integers(0, n_docs, size=(B, W)), r.integers(1, n_docs, size=(B, K1)))))
    return out


def _state(model):
    names = ['dense_param', 'dense_grad', 'dense_m', 'dense_v', 'user_emb', 'user_m', 'user_v']
    if model.arch in ('iigru', 'iicgru'):
        names += ['user_emb2', 'user_m2', 'user_v2']
    return {n: getattr(model, n).cpu().numpy() for n in names}


@pytest.mark.parametrize('arch,dims', [('igru', MID), ('gru', MID), ('hgru', MID), ('iigru', MID), ('nigru', MID), ('vo', MID),
                                       ('igru', ODD), ('gru', ODD)])
def test_fused_step_is_bit_identical_to_split_calls(arch, dims):
    """(ODD: k = 5 taps, F and U off the tile sizes -- the weight images the fused step writes from inside its Adam launch
    have padding the stand-alone pack kernels wrote once; the predictions after four steps go through those images.)"""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    B, W, T, K1 = 24, 7, 11, 5
    outs = []
    for mode in ('fused', 'split'):
        O, p, model, rng = build(arch, dims, B, W, T, K1, 0.2)
        steps = _batches(model, B, W, K1, T, 4, dims['N_u'], vocab=dims['V'])
        losses = []
        for i, (u, h, c) in enumerate(steps):
            if mode == 'fused':
                model._train_step(u, h, c, True)
            else:
                s = model._stream()
                check(lib.lstur_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, i, 1.0,
                                                     model.stats.data_ptr(), s), 'fb')
                check(lib.lstur_adam_step(model._ctx, 1e-3, i + 1, None, None, 0, s), 'adam')
            losses.append(model.stats[0].item())
        torch.cuda.synchronize()
        st = _state(model)
        st['loss'] = np.array(losses)
        # the images packed behind the last Adam are the ones a forward right now must see
        st['pred'] = model.predict_ids(*steps[0])
        outs.append(st)
        del model
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert np.isfinite(outs[0]['loss']).all() and len(set(outs[0]['loss'])) > 1


def test_fused_pointwise_step_is_bit_identical_to_split_calls():
    _need_gpu()
    from mnexp_amd._lib import lib, check
    from test_gpu_pointwise import build as pbuild
    B, W, T, K = 19, 6, 9, 4
    outs = []
    for mode in ('fused', 'split'):
        O, p, model, rng = pbuild('iicgru', 'point_mlp', MID, B, W, T, K, 0.2, 1.3)
        steps = _batches(model, B, W, 1, T, 3, MID['N_u'])
        labels = [model._dev_f32((np.random.default_rng(9 + i).random(B) < 0.3).astype(np.float32)) for i in range(3)]
        for i, (u, h, c) in enumerate(steps):
            c = c.reshape(-1).contiguous()
            if mode == 'fused':
                model._train_step(u, h, c, True, labels[i])
            else:
                s = model._stream()
                check(lib.lstur_point_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(),
                                                           labels[i].data_ptr(), B, i, 1.0, model.stats.data_ptr(), s), 'fb')
                check(lib.lstur_adam_step(model._ctx, 1e-3, i + 1, None, None, 0, s), 'adam')
        torch.cuda.synchronize()
        outs.append(_state(model))
        del model
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_set_weights_between_steps_invalidates_the_weight_images():
    """The library packs the conv / GRU weight images behind every Adam step; parameters written from outside
    (set_weights -> lstur_params_changed) must reach the next forward."""
    _need_gpu()
    B, W, T, K1 = 16, 5, 9, 5
    O, p, model, rng = build('igru', MID, B, W, T, K1, 0.0)
    steps = _batches(model, B, W, K1, T, 2, MID['N_u'])
    model._train_step(*steps[0], True)
    other = {n: a for n, a in p.items() if n != 'word_emb'}
    r = np.random.default_rng(4)
    other['conv_w'] = r.normal(0, 0.05, size=p['conv_w'].shape)
    other['gru_u'] = r.normal(0, 0.05, size=p['gru_u'].shape)
    model.set_weights(other)
    u, h, c = steps[1]
    got = model.predict([u.cpu().numpy(), model.title_table[h.long()].cpu().numpy(),
                         model.title_table[c[:, 0].long()].cpu().numpy()])
    O2, p2, fresh, _ = build('igru', MID, B, W, T, K1, 0.0)
    fresh.set_weights(other)
    fresh.set_title_table(model.title_table.cpu().numpy())
    want = fresh.predict([u.cpu().numpy(), model.title_table[h.long()].cpu().numpy(),
                          model.title_table[c[:, 0].long()].cpu().numpy()])
    assert np.array_equal(got, want)


@pytest.mark.parametrize('arch', ['igru', 'iigru'])
def test_deferred_table_sweep_of_the_exchanged_row_step_is_bit_identical(arch):
    """lstur_adam_step with EXTERNAL rows (the data-parallel step: rows and gradients of every rank's batch) updates those
    rows at once and lets the zero-gradient sweep of every other row wait for the next forward, under whose recurrence it
    runs.  Six steps with overlapping users, an evaluation in between (the pending sweep then runs in
    the open) and a final flush against the same steps with the sweep inside lstur_adam_step (rows = NULL: no deferral):
    tables, moments, parameters, losses and predictions bit-identical."""
    _need_gpu()
    import ctypes as C
    from mnexp_amd._lib import lib, check
    B, W, T, K1 = 24, 7, 11, 5
    outs = []
    for mode in ('external', 'plain'):
        O, p, model, rng = build(arch, MID, B, W, T, K1, 0.2)
        steps = _batches(model, B, W, K1, T, 6, MID['N_u'])
        losses, preds = [], []
        for i, (u, h, c) in enumerate(steps):
            s = model._stream()
            check(lib.lstur_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, i, 1.0,
                                                 model.stats.data_ptr(), s), 'fb')
            if mode == 'external':
                rows, grads, cnt = C.c_void_p(), C.c_void_p(), C.c_int32()
                if arch == 'iigru':      # both tables' gradients travel through the exchange buffers
                    world = 1
                    used = int(lib.lstur_dp_tail_floats(C.byref(model.cfg), world))
                    tail = model._bucket[model.sizes.dense:model.sizes.dense + used]
                    check(lib.lstur_dp_stage(model._ctx, tail.data_ptr(), 0, world, s), 'stage')
                    check(lib.lstur_dp_unpack(model._ctx, tail.data_ptr(), world, B, C.byref(rows), C.byref(grads), s), 'unpack')
                    n_rows = B
                else:
                    check(lib.lstur_user_grad(model._ctx, C.byref(rows), C.byref(grads), C.byref(cnt)), 'user_grad')
                    n_rows = cnt.value
                check(lib.lstur_adam_step(model._ctx, 1e-3, i + 1, rows, grads, n_rows, s), 'adam')
            else:
                check(lib.lstur_adam_step(model._ctx, 1e-3, i + 1, None, None, 0, s), 'adam')
            losses.append(model.stats[0].item())
            if i == 2:
                preds.append(model.predict_ids(*steps[0]))     # reads the tables: a pending sweep runs first, in the open
        model.flush_tables()
        torch.cuda.synchronize()
        st = _state(model)
        st['loss'] = np.array(losses)
        st['pred_mid'] = preds[0]
        st['pred'] = model.predict_ids(*steps[0])
        outs.append(st)
        del model
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


_SPLIT_CHILD = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from test_gpu_model import MID, build
from test_gpu_train_step import _batches, _state
B, W, T, K1 = 24, 7, 11, 5
O, p, model, rng = build(sys.argv[2], MID, B, W, T, K1, 0.2)
steps = _batches(model, B, W, K1, T, 4, MID['N_u'])
for u, h, c in steps:
    model._train_step(u, h, c, True)
torch.cuda.synchronize()
model.flush_tables()
st = _state(model)
st['pred'] = model.predict_ids(*steps[0])
np.savez(sys.argv[1], **st)
'''


@pytest.mark.parametrize('arch', ['igru', 'gru'])
def test_candidate_split_of_the_encoder_backward_changes_only_the_summation_order(arch, tmp_path):
    """The candidate titles' encoder backward on the side stream under BPTT (default) against one launch per
    kernel over all titles (LSTUR_CAND_SPLIT=0): the same products, reduced in a different slab order -- parameters, moments,
    tables and predictions after four steps agree to 2e-4 of scale (1e-2 for the Adam moments of near-zero gradients)."""
    _need_gpu()
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, env in (('split', {}), ('whole', {'LSTUR_CAND_SPLIT': '0'})):
        f = str(tmp_path / (tag + '.npz'))
        subprocess.check_call([sys.executable, '-c', _SPLIT_CHILD % (root, os.path.join(root, 'tests')), f, arch],
                              env=dict(os.environ, **env), cwd=root)
        outs.append(np.load(f))
    for k in outs[0].files:
        a, b = outs[0][k].astype(np.float64), outs[1][k].astype(np.float64)
        scale = max(np.abs(b).max(), 1e-6)
        tol = 1e-2 if k.endswith('_m') or k.endswith('_v') or k in ('dense_m', 'dense_v') else 2e-4
        assert np.abs(a - b).max() <= tol * scale + 1e-7, (k, np.abs(a - b).max(), scale)
