"""Size-independent properties of the HIP path at the FULL shapes of BASELINE.json (configs[1]: B = 64 and the per-GPU
shape of configs[3]: B = 512, 1 M users / 200 k docs), where an oracle run takes minutes: what must hold whatever the
numbers are.
  * the order of the samples in a batch changes nothing but the grouping of one fp32 sum: every title, every recurrence row
    and every score is computed from its own inputs only (reference: the graph of task/paper.py:584-665 has no cross-sample
    operation; tiles, slabs and workgroups of the kernels do mix samples, which is what this pins).  Permuting the batch
    permutes the logits to 2e-6 of their scale -- not bit for bit: the conv forward cuts the row tiles of its last partial
    round into k-slices (convx.hip), so WHERE a token sits in the compacted list decides whether its 912-term dot products
    are summed whole or as up to eight partial sums;
  * the loss gradient is linear in the loss scale, and exactly so for a power of two: forward_backward(grad_scale = 1/2)
    leaves exactly half of every gradient (the fp16-pair kernel gradient re-derives its power-of-two scale from max |dC|,
    so halving dC shifts exponents and nothing else);
  * the same step from the same state twice gives the same bits (fixed-order reductions everywhere: no atomics on floats);
  * a ragged last batch (n < max_batch) through the in-place data-parallel optimizer entry point equals the staged /
    unpacked sequence bit for bit."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _model(name, dropout, seed=7, n_batches=2):
    import bench
    from mnexp_amd.native_model import NativeModel
    wl = bench.WORKLOADS[name]
    titles, word, batches = bench.synth_inputs(wl, np.random.default_rng(99), n_batches)
    model = NativeModel(word, wl['N_u'], wl['B'], wl['W'], wl['T'], wl['K1'] - 1, wl['F'], wl['k'], wl['U'], arch=wl['arch'],
                        dropout=dropout, lr=1e-3, seed=seed, title_table=titles)
    return wl, model, titles, batches


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')


@pytest.mark.parametrize('name', ['lstur_ini_b64', 'lstur_con_b64', 'lstur_ini_b512_1Musers'])
def test_permuting_the_batch_permutes_the_logits(name):
    _need_gpu()
    from mnexp_amd._lib import lib, check
    wl, model, titles, batches = _model(name, 0.0)
    B, K1 = wl['B'], wl['K1']
    user, hist, cands = batches[0]
    perm = np.random.default_rng(3).permutation(B)

    def logits(u, h, c):
        ud, hd, cd = (model._dev_i32(a) for a in (u, h, c))
        check(lib.lstur_forward_backward_ids(model._ctx, ud.data_ptr(), hd.data_ptr(), cd.data_ptr(), B, 0, 1.0,
                                             model.stats.data_ptr(), model._stream()), 'fb')
        torch.cuda.synchronize()
        return model.peek('logits').cpu().numpy().reshape(B, K1).copy()

    a = logits(user, hist, cands)
    b = logits(user[perm], hist[perm], cands[perm])
    assert np.isfinite(a).all() and np.abs(a).max() > 0
    err = np.abs(a[perm].astype(np.float64) - b).max() / np.abs(a).max()
    assert err <= 2e-6, err
    # ... and the user-row gradients come back in the batch's order
    g = model.get_grads()
    assert np.array_equal(g['user_rows'], user[perm])


@pytest.mark.parametrize('name', ['lstur_ini_b64', 'lstur_con_b64'])
def test_gradients_are_exactly_linear_in_a_power_of_two_loss_scale(name):
    _need_gpu()
    from mnexp_amd._lib import lib, check
    wl, model, titles, batches = _model(name, 0.2)
    B = wl['B']
    u, h, c = (model._dev_i32(a) for a in batches[0])

    def grads(scale):
        check(lib.lstur_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 5, scale,
                                             model.stats.data_ptr(), model._stream()), 'fb')
        torch.cuda.synchronize()
        return {k: np.array(v, copy=True) for k, v in model.get_grads().items()}

    g1, gh = grads(1.0), grads(0.5)
    for k in g1:
        if k == 'user_rows':
            assert np.array_equal(g1[k], gh[k])
            continue
        a, b = np.asarray(g1[k], np.float32), np.asarray(gh[k], np.float32)
        assert np.abs(a).max() > 0, k
        # exact halves wherever the half is a normal float (a gradient element down at 1e-38 may lose its last bit)
        big = np.abs(a) > 1e-30
        assert np.array_equal((a[big] * np.float32(0.5)).view(np.uint32), b[big].view(np.uint32)), k


@pytest.mark.parametrize('name', ['lstur_ini_b64'])
def test_the_same_steps_from_the_same_state_give_the_same_bits(name):
    _need_gpu()
    out = []
    for _ in range(2):
        wl, model, titles, batches = _model(name, 0.2, n_batches=3)
        dev = [tuple(model._dev_i32(a) for a in b) for b in batches]
        for i, d in enumerate(dev):
            if i + 1 < len(dev):
                model.prefetch_ids(dev[i + 1][1], dev[i + 1][2], dev[i + 1][0])
            model._train_step(*d, True)
        torch.cuda.synchronize()
        w = model.get_weights()
        out.append({k: np.array(v, copy=True) for k, v in w.items()})
        del model
    for k in out[0]:
        assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), k


@pytest.mark.parametrize('arch', ['igru', 'iigru'])
@pytest.mark.parametrize('n', [5, 8])
def test_in_place_dp_optimizer_step_equals_staged_and_unpacked_rows(arch, n):
    """lstur_dp_attach + lstur_adam_step_dp (the tail staged by the backward pass's reduction launch, the reduced tail read
    in place) against lstur_dp_stage + lstur_dp_unpack + lstur_adam_step on a second context: same parameters, tables and
    moments bit for bit, for a full batch (n == max_batch: the in-place path) and a ragged one (n < max_batch: unpacked
    internally).  Two emulated ranks: the other rank's slot is filled by a second pair of contexts."""
    _need_gpu()
    from oracle import lstur_oracle as O
    from mnexp_amd._lib import lib, check
    from mnexp_amd.native_model import NativeModel
    dims = dict(V=120, D=24, F=32, k=3, U=16, N_u=40)
    B, W, T, K1, world = 8, 9, 10, 3, 2
    rng = np.random.default_rng(11)
    p = O.init_params(rng, arch=arch, dtype=np.float32, **dims)
    batches = []
    for r in range(world):
        user = rng.integers(0, dims['N_u'], size=n).astype(np.int32)
        user[0] = 3                                   # the same user on both ranks: the optimizer sums the duplicates
        hist = rng.integers(0, dims['V'], size=(n, W, T)).astype(np.int32)
        cands = rng.integers(1, dims['V'], size=(n, K1, T)).astype(np.int32)
        batches.append((user, hist, cands))

    def make():
        m = NativeModel(p['word_emb'], dims['N_u'], B, W, T, K1 - 1, dims['F'], dims['k'], dims['U'], arch=arch, dropout=0.2,
                        lr=1e-2, seed=3)
        m.set_weights({k: v for k, v in p.items() if k != 'word_emb'})
        return m

    results = []
    for fold in (True, False):
        ms = [make() for _ in range(world)]
        dense = ms[0].sizes.dense
        used = int(lib.lstur_dp_tail_floats(C.byref(ms[0].cfg), world))
        for r, m in enumerate(ms):
            if fold:
                check(lib.lstur_dp_attach(m._ctx, m._bucket[dense:].data_ptr(), r, world), 'attach')
            u, h, c = (m._dev_i32(a) for a in batches[r])
            check(lib.lstur_forward_backward(m._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), n, 0, 1.0 / world,
                                             m.stats.data_ptr(), m._stream()), 'fb')
            if not fold:
                check(lib.lstur_dp_stage(m._ctx, m._bucket[dense:].data_ptr(), r, world, m._stream()), 'stage')
        torch.cuda.synchronize()
        total = sum(m._bucket[:dense + used].clone() for m in ms)      # the all-reduce
        for m in ms:
            m._bucket[:dense + used].copy_(total)
            if fold:
                check(lib.lstur_adam_step_dp(m._ctx, 1e-2, 1, m._bucket[dense:].data_ptr(), world, n, m._stream()), 'adam_dp')
            else:
                rows, grads = C.c_void_p(), C.c_void_p()
                check(lib.lstur_dp_unpack(m._ctx, m._bucket[dense:].data_ptr(), world, n, C.byref(rows), C.byref(grads),
                                          m._stream()), 'unpack')
                check(lib.lstur_adam_step(m._ctx, 1e-2, 1, rows.value, grads.value, world * n, m._stream()), 'adam')
            check(lib.lstur_flush_tables(m._ctx, m._stream()), 'flush')
        torch.cuda.synchronize()
        ws = [{k: np.array(v, copy=True) for k, v in m.get_weights().items()} for m in ms]
        for k in ws[0]:                                                  # replicas agree
            assert np.array_equal(ws[0][k].view(np.uint32), ws[1][k].view(np.uint32)), (fold, k)
        results.append(ws[0])
    moved = 0.0
    for k in results[0]:
        assert np.array_equal(results[0][k].view(np.uint32), results[1][k].view(np.uint32)), k
        moved = max(moved, float(np.abs(results[0][k].astype(np.float64) - np.asarray(p[k], np.float64)).max()))
    assert moved > 1e-3
