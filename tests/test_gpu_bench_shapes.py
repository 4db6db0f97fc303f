"""Parity at the EXACT shapes bench.py measures (BASELINE.json configs[1], [2] and the build-defined mhsa encoder at the
config-2 shapes: B = 64, W = 50, T = 30, 1 + 4 candidates, D = 300, 50 k users / 20 k docs / 50 k vocab; batches from
bench.synth_inputs): logits, every dense gradient and the user-row gradients of one step, then two fused Adam steps
(forward + backward + optimizer in ONE call, the way bench.py and fit_generator drive the library) against
oracle.train_step in fp64 (the parameters are fp32 values).  The oracle runs in fp64 on purpose: at these sizes a conv output
that lies within 1e-7 of zero changes the side of the ReLU it falls on, and an fp32 evaluation (the fp32 oracle and the
exact-fp32 MFMA kernels alike: tools/diag_bench_shapes.py) then differs from exact math by 3e-3 .. 6e-3 of scale in conv_w /
conv_b -- the split-operand kernels the step runs by default are within 1e-6.  The conv kernel gradient's slab reduction, the candidate / history split launches and BPTT over
16 workgroups x 50 steps at these sizes are otherwise only checked against themselves.  mhsa at the config-4 per-GPU shape
(B = 512, 1 M users) is checked forward-only like the cnnatt test of tests/test_gpu_kernels.py.
Reference: task/paper.py:584-665 (cnnatt path); the mhsa encoder has no reference counterpart: parity is
against the repo's own bf16-emulating oracle."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from test_gpu_model import rel_close  # noqa: E402


def _setup(name, dropout, seed=7, n_batches=3):
    import bench
    from oracle import lstur_oracle as O
    from mnexp_amd.native_model import NativeModel
    wl = bench.WORKLOADS[name]
    titles, word, batches = bench.synth_inputs(wl, np.random.default_rng(1234), n_batches)
    enc = wl.get('encoder', 'cnnatt')
    rng = np.random.default_rng(5)
    p = O.init_params(rng, V=wl['V'], D=wl['D'], F=wl['F'], k=wl['k'], U=wl['U'], N_u=wl['N_u'], arch=wl['arch'],
                      word_emb=word, dtype=np.float32, encoder=enc, heads=wl.get('heads', 1))
    for n in (('qkv_b' if enc == 'mhsa' else 'conv_b'), 'att_b', 'dense_b', 'gru_b'):
        p[n] = rng.normal(0, 0.1, size=p[n].shape).astype(np.float32)
    # fp32 VALUES, fp64 arithmetic in the oracle
    p = {n: (a.astype(np.float64) if isinstance(a, np.ndarray) and a.dtype.kind == 'f' else a) for n, a in p.items()}
    model = NativeModel(word, wl['N_u'], wl['B'], wl['W'], wl['T'], wl['K1'] - 1, wl['F'], wl['k'], wl['U'], arch=wl['arch'],
                        dropout=dropout, lr=1e-3, seed=seed, title_table=titles, news_encoder=enc, heads=wl.get('heads', 0))
    model.set_weights({n: p[n] for n in p if n not in ('word_emb', 'mhsa_heads')})
    return O, wl, p, model, titles, batches


@pytest.mark.parametrize('name', ['lstur_ini_b64', 'lstur_con_b64', 'mhsa_b64'])
def test_one_step_and_two_adam_steps_at_the_bench_shapes(name):
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    from mnexp_amd._lib import lib, check
    O, wl, p, model, titles, batches = _setup(name, 0.2)
    arch, enc, B, K1 = wl['arch'], wl.get('encoder', 'cnnatt'), wl['B'], wl['K1']
    mh = enc == 'mhsa'
    assert model.conv_scheme == ('n/a' if mh else 'f16x2')
    tol_l, tol_g = (2e-3, 8e-3) if mh else (1e-3, 2e-3)      # (mhsa: bf16 MFMA operands, tests/test_gpu_mhsa.py)
    user, hist, cands = batches[0]
    u, h, c = (model._dev_i32(a) for a in (user, hist, cands))
    # ---- one forward + backward (split call: the gradients stay in the bucket), dropout on (identical masks: counter hash)
    ref = O.forward_backward(p, arch, user, titles[hist], titles[cands], drop_p=0.2, seed=7, step=0)
    check(lib.lstur_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 0, 1.0,
                                         model.stats.data_ptr(), model._stream()), 'fb')
    torch.cuda.synchronize()
    logits = model.peek('logits').cpu().numpy().reshape(B, K1)
    rel_close(logits, ref['logits'], tol_l, 'logits')
    assert abs(model.stats[0].item() - ref['loss']) <= 1e-3 * max(1.0, abs(ref['loss']))
    g = model.get_grads()
    for n in O.dense_param_names(arch, encoder=enc):
        rel_close(g[n], ref['grads'][n], 2e-2 if n == 'qkv_b' else tol_g, n)
    assert np.array_equal(g['user_rows'], user)
    rel_close(g['user_row_grads'], ref['user_row_grads'], tol_g, 'user_row_grads')
    # ---- two fused training steps from the same start (doc-id batches, the next batch announced: prefetched staging)
    model.set_weights({n: p[n] for n in p if n not in ('word_emb', 'mhsa_heads')})   # (also resets nothing else: m = v = 0 still)
    pf = {n: (a.copy() if isinstance(a, np.ndarray) else a) for n, a in p.items()}
    opt = O.Adam(pf, O.dense_param_names(arch, encoder=enc) + ['user_emb'], lr=1e-3)
    dev = [tuple(model._dev_i32(a) for a in b) for b in batches[:2]]
    for i, (user, hist, cands) in enumerate(batches[:2]):
        r = O.train_step(pf, arch, opt, user, titles[hist], titles[cands], drop_p=0.2, seed=7, step=i)
        if i == 0:
            model.prefetch_ids(dev[1][1], dev[1][2], dev[1][0])
        model._train_step(*dev[i], True)
        torch.cuda.synchronize()
        assert abs(model.stats[0].item() - r['loss']) <= 2e-3 * max(1.0, abs(r['loss'])), (i, model.stats[0].item(), r['loss'])
    if not mh:
        assert lib.lstur_prefetch_hits(model._ctx) == 1
    w = model.get_weights()
    # Adam turns a gradient into a step of about lr whatever its size: an element whose gradient is ~1e-8 (most of conv_w's
    # at these shapes) moves by lr * m / (sqrt(v) + eps) with eps = 1e-7 in the denominator's range, so a 1e-9 gradient
    # difference shows as a fraction of a learning-rate step.  Bound: 2 lr absolute (a tenth of that is typical) + 1e-2 of scale
    for n in O.dense_param_names(arch, encoder=enc) + ['user_emb']:
        a, b = w[n].reshape(-1).astype(np.float64), pf[n].reshape(-1).astype(np.float64)
        err = np.abs(a - b).max()
        assert err <= 2.2e-3 and np.abs(a - b).mean() <= 2e-4, (n, err, np.abs(a - b).mean())
        moved = np.abs(b - np.asarray(p[n], np.float64).reshape(-1))
        assert moved.max() > 1e-4, n                         # the parameter was trained at all


def test_mhsa_forward_at_config4_shapes_batch_512_one_million_users():
    """BASELINE.json configs[4] per GPU: the build-defined mhsa news encoder at B = 512, 1 M users (bench workload
    mhsa_b512_1Musers): logits and loss of a full batch against a forward-only oracle call, as
    test_config4_shapes_batch_512_one_million_users does for cnnatt.  Tolerance: 2e-3 of scale (bf16 MFMA operands)."""
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    import bench
    from oracle import lstur_oracle as O
    from mnexp_amd.native_model import NativeModel
    from mnexp_amd._lib import lib, check
    wl = dict(bench.WORKLOADS['mhsa_b512_1Musers'], V=3000, N_d=4000)     # (word / title tables cut down: the oracle gathers them)
    titles, word, batches = bench.synth_inputs(wl, np.random.default_rng(3), 1)
    rng = np.random.default_rng(5)
    p = O.init_params(rng, V=wl['V'], D=wl['D'], F=wl['F'], k=1, U=wl['U'], N_u=wl['N_u'], arch='igru', word_emb=word,
                      dtype=np.float32, encoder='mhsa', heads=wl['heads'])
    model = NativeModel(word, wl['N_u'], wl['B'], wl['W'], wl['T'], wl['K1'] - 1, wl['F'], 1, wl['U'], arch='igru', dropout=0.0,
                        lr=1e-3, seed=7, title_table=titles, news_encoder='mhsa', heads=wl['heads'])
    assert model.lazy_table
    model.set_weights({n: p[n] for n in p if n not in ('word_emb', 'mhsa_heads')})
    user, hist, cands = batches[0]
    user[:4] = [0, 999999, 123456, 999999]
    ref = O.forward_backward(p, 'igru', user, titles[hist], titles[cands], drop_p=0.0, want_grads=False)
    u, h, c = (model._dev_i32(a) for a in (user, hist, cands))
    check(lib.lstur_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), wl['B'], 0, 1.0,
                                         model.stats.data_ptr(), model._stream()), 'fb')
    torch.cuda.synchronize()
    logits = model.peek('logits').cpu().numpy().reshape(wl['B'], wl['K1'])
    rel_close(logits, ref['logits'], 2e-3, 'logits')
    assert abs(model.stats[0].item() - ref['loss']) <= 2e-3 * max(1.0, abs(ref['loss']))
    g = model.get_grads()
    assert np.array_equal(g['user_rows'], user) and np.isfinite(g['user_row_grads']).all()
