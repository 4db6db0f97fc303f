"""Per-kernel GPU parity through the C-ABI entry points (include/lstur.h), at the BASELINE.json shapes
(D=300, T=30, F=400, k=3, U=200) where the specialised kernels (25-tile conv forward, 15-wave kernel
gradient) are selected, plus bit-exact checks of the integer/mask work."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def stream():
    return torch.cuda.current_stream().cuda_stream


def pack_b(B):
    from mnexp_amd._lib import lib, check
    K, N = B.shape
    out = torch.zeros(lib.lstur_pack_b_floats(K, N), dtype=torch.float32, device='cuda')
    check(lib.lstur_pack_b(B.data_ptr(), B.stride(0), K, N, out.data_ptr(), stream()))
    return out


def test_gather_is_bit_exact_with_dropout_and_padding():
    _need_gpu()
    from mnexp_amd._lib import lib, check
    from oracle import lstur_oracle as O
    rng = np.random.default_rng(0)
    V, D, N, T, k = 5000, 300, 200, 30, 3
    table = rng.normal(size=(V, D)).astype(np.float32)
    ids = rng.integers(0, V, size=(N, T)).astype(np.int32)
    ids[:, 20:] = 0
    for drop in (0.0, 0.2):
        key = O.dropout_key(9, 4, 0)
        out = torch.full((N, T + 2, D), 7.0, dtype=torch.float32, device='cuda')
        table_d, ids_d = dev(table), dev(ids)      # keep operands alive: a freed temporary's block is reused at once
        check(lib.lstur_gather_f32(table_d.data_ptr(), ids_d.data_ptr(), out.data_ptr(), N, T, D, 1, 1, drop, key, stream()))
        ref = table[ids]
        if drop:
            keep = O.dropout_keep(key, N * T * D, drop).reshape(N, T, D)
            ref = ref * keep * np.float32(1.0 / (1.0 - drop))
        got = out.cpu().numpy()
        assert np.array_equal(got[:, 1:-1], ref.astype(np.float32))
        assert not got[:, 0].any() and not got[:, -1].any()


@pytest.mark.parametrize('N', [3, 130])
def test_conv_fwd_and_wgrad_at_baseline_shapes(N):
    _need_gpu()
    from mnexp_amd._lib import lib, check
    rng = np.random.default_rng(1)
    D, T, F, k = 300, 30, 400, 3
    E = rng.normal(0, 0.3, size=(N, T, D)).astype(np.float32)
    ids = rng.integers(0, 50, size=(N, T)).astype(np.int32)
    ids[:, 25:] = 0
    Wc = rng.normal(0, 0.05, size=(k, D, F)).astype(np.float32)
    b = rng.normal(0, 0.1, size=F).astype(np.float32)
    Ep = np.zeros((N, T + 2, D), np.float32)
    Ep[:, 1:-1] = E
    Epd = torch.zeros(N * (T + 2) * D + 256, dtype=torch.float32, device='cuda')
    Epd[:N * (T + 2) * D] = dev(Ep).view(-1)
    Wd = dev(Wc.reshape(k * D, F))
    Wp = pack_b(Wd)
    out = torch.empty(N * T, F, dtype=torch.float32, device='cuda')
    b_d, ids_d = dev(b), dev(ids)
    check(lib.lstur_conv1d_relu_mask_fwd(Epd.data_ptr(), Wp.data_ptr(), b_d.data_ptr(), ids_d.data_ptr(),
                                         out.data_ptr(), N, T, D, k, F, None, None, stream()))
    x = torch.nn.functional.pad(dev(E).double().transpose(1, 2), (1, 1))
    ref = torch.relu(torch.nn.functional.conv1d(x, dev(Wc).double().permute(2, 1, 0), dev(b).double()))
    ref = ref.transpose(1, 2) * (dev(ids) != 0).unsqueeze(-1)
    err = (out.view(N, T, F).double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
    assert (out.view(N, T, F)[dev(ids) == 0] == 0).all()

    dC = rng.normal(size=(N * T, F)).astype(np.float32)
    dw = torch.empty(k * D, F, dtype=torch.float32, device='cuda')
    scratch = torch.empty(lib.lstur_conv1d_wgrad_scratch(D, k, F), dtype=torch.uint8, device='cuda')
    dC_d = dev(dC)
    check(lib.lstur_conv1d_wgrad(Epd.data_ptr(), dC_d.data_ptr(), dw.data_ptr(), scratch.data_ptr(), N, T, D, k, F,
                                 None, None, stream()))
    Ep_d = dev(Ep)
    cols = torch.stack([Ep_d[:, j:j + T] for j in range(k)], dim=2).reshape(N * T, k * D).double()
    ref_dw = cols.t() @ dC_d.double()
    err = (dw.double() - ref_dw).abs().max().item()
    assert err < 2e-5 * ref_dw.abs().max().item(), err

    # --- the same two contractions over the compacted list of non-pad tokens (what the training step runs)
    vidx = torch.full((N * T,), -1, dtype=torch.int32, device='cuda')
    vcount = torch.zeros(1, dtype=torch.int32, device='cuda')
    vs = torch.zeros(max(1, lib.lstur_valid_tokens_scratch(N * T) // 4), dtype=torch.int32, device='cuda')
    check(lib.lstur_valid_tokens(ids_d.data_ptr(), N * T, vidx.data_ptr(), vcount.data_ptr(), vs.data_ptr(), stream()))
    want = np.flatnonzero(ids.reshape(-1)).astype(np.int32)
    assert int(vcount.item()) == len(want) and np.array_equal(vidx[:len(want)].cpu().numpy(), want)   # bit-exact, stable
    out2 = torch.full((N * T, F), -7.0, dtype=torch.float32, device='cuda')
    check(lib.lstur_conv1d_relu_mask_fwd(Epd.data_ptr(), Wp.data_ptr(), b_d.data_ptr(), ids_d.data_ptr(),
                                         out2.data_ptr(), N, T, D, k, F, vidx.data_ptr(), vcount.data_ptr(), stream()))
    valid = dev(ids.reshape(-1) != 0)
    assert torch.equal(out2[valid], out[valid])              # same rows, bit-identical
    assert (out2[~valid] == -7.0).all()                       # pad rows untouched
    dC_m = dC_d * valid.unsqueeze(-1)                         # true gradients are zero on pad rows
    dw_ref2 = torch.empty_like(dw)
    check(lib.lstur_conv1d_wgrad(Epd.data_ptr(), dC_m.data_ptr(), dw_ref2.data_ptr(), scratch.data_ptr(), N, T, D, k, F,
                                 None, None, stream()))
    dC_g = torch.where(valid.unsqueeze(-1), dC_d, torch.full_like(dC_d, float('nan')))   # pad rows must not be read
    dw2 = torch.empty_like(dw)
    check(lib.lstur_conv1d_wgrad(Epd.data_ptr(), dC_g.data_ptr(), dw2.data_ptr(), scratch.data_ptr(), N, T, D, k, F,
                                 vidx.data_ptr(), vcount.data_ptr(), stream()))
    err = (dw2.double() - dw_ref2.double()).abs().max().item()
    assert err < 2e-5 * ref_dw.abs().max().item(), err


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(3520, 200, 400), (200, 600, 3200), (37, 19, 70), (64, 64, 2048), (3200, 600, 200),
                                   (100, 52, 332), (400, 200, 3520), (8, 12, 260)])
def test_gemm_variants(ta, tb, M, N, K):
    _need_gpu()
    from mnexp_amd._lib import lib, check
    g = torch.Generator(device='cuda').manual_seed(0)
    A = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g)
    B = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g)
    bias = torch.randn(N, device='cuda', generator=g)
    Cm = torch.empty(M, N, device='cuda')
    scratch = torch.empty(max(1, lib.lstur_gemm_scratch(M, N, K)), dtype=torch.uint8, device='cuda')
    check(lib.lstur_gemm_f32(A.data_ptr(), A.stride(0), ta, B.data_ptr(), B.stride(0), tb, Cm.data_ptr(), N,
                             bias.data_ptr(), M, N, K, 0.5, scratch.data_ptr(), stream()))
    ref = 0.5 * ((A.t() if ta else A).double() @ (B.t() if tb else B).double()) + bias.double()
    err = (Cm.double() - ref).abs().max().item()
    assert err < 1e-5 * ref.abs().max().item() + 1e-6, err


@pytest.mark.parametrize('b_nk', [0, 1])
@pytest.mark.parametrize('M,N,K', [(3520, 200, 400), (3200, 600, 200), (3200, 200, 600), (3520, 400, 200), (37, 20, 72),
                                   (64, 64, 2048), (100, 52, 332), (8, 12, 260), (130, 70, 4)])
def test_gemm_bf16x3_is_fp32_grade(b_nk, M, N, K):
    """csrc/gemm3.hip (the Dense(U) / GRU input projections on the critical path): three bf16 parts per operand, six
    products -- held to the exact-fp32 kernel's bound against fp64, with the k tail inside a 32-wide step, ragged tiles
    and operands embedded in wider rows (lda / ldb > K)."""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    g = torch.Generator(device='cuda').manual_seed(1)
    lda = K + 8
    Afull = torch.randn(M, lda, device='cuda', generator=g)
    Bfull = torch.randn((N, K + 4) if b_nk else (K, (N + 15) // 4 * 4), device='cuda', generator=g)   # ld % 4 == 0
    A, B = Afull[:, :K], (Bfull[:, :K] if b_nk else Bfull[:, :N])
    bias = torch.randn(N, device='cuda', generator=g)
    ldc = N + 3
    Cm = torch.full((M, ldc), float('nan'), device='cuda')
    check(lib.lstur_gemm_bf16x3(Afull.data_ptr(), lda, Bfull.data_ptr(), Bfull.stride(0), b_nk, Cm.data_ptr(), ldc,
                                bias.data_ptr(), M, N, K, stream()))
    torch.cuda.synchronize()
    ref = A.double() @ (B.t() if b_nk else B).double() + bias.double()
    err = (Cm[:, :N].double() - ref).abs().max().item()
    assert err < 1e-5 * ref.abs().max().item() + 1e-6, err
    assert torch.isnan(Cm[:, N:]).all()                       # nothing written past column N
    # and against the exact-fp32 MFMA kernel: a few ulp
    C2 = torch.empty(M, N, device='cuda')
    scratch = torch.empty(max(1, lib.lstur_gemm_scratch(M, N, K)), dtype=torch.uint8, device='cuda')
    check(lib.lstur_gemm_f32(Afull.data_ptr(), lda, 0, Bfull.data_ptr(), Bfull.stride(0), b_nk, C2.data_ptr(), N,
                             bias.data_ptr(), M, N, K, 1.0, scratch.data_ptr(), stream()))
    assert (Cm[:, :N] - C2).abs().max().item() < 4e-6 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize('M,N,K', [(200, 600, 3200), (200, 400, 3200), (200, 200, 3200), (400, 200, 3520), (70, 36, 1000),
                                   (8, 12, 260), (130, 68, 5), (64, 64, 128)])
def test_gemm_bf16x3_tn_is_fp32_grade(M, N, K):
    """The parameter-gradient form (A^T . B over tokens / history slots, K split into slabs): GRU kernel / recurrent /
    Dense(U) weight gradients at the baseline shapes, plus ragged tiles, a k tail inside a step and K below one step."""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    g = torch.Generator(device='cuda').manual_seed(2)
    lda, ldb, ldc = (M + 11) // 4 * 4, (N + 7) // 4 * 4, N + 5
    Afull = torch.randn(K, lda, device='cuda', generator=g)
    Bfull = torch.randn(K, ldb, device='cuda', generator=g)
    Cm = torch.full((M, ldc), float('nan'), device='cuda')
    scratch = torch.empty(max(1, lib.lstur_gemm_bf16x3_tn_scratch(M, N, K)), dtype=torch.uint8, device='cuda')
    check(lib.lstur_gemm_bf16x3_tn(Afull.data_ptr(), lda, Bfull.data_ptr(), ldb, Cm.data_ptr(), ldc, M, N, K,
                                   scratch.data_ptr(), stream()))
    torch.cuda.synchronize()
    ref = Afull[:, :M].double().t() @ Bfull[:, :N].double()
    err = (Cm[:, :N].double() - ref).abs().max().item()
    assert err < 1e-5 * ref.abs().max().item() + 1e-6, err
    assert torch.isnan(Cm[:, N:]).all()


@pytest.mark.parametrize('generic', [False, True], ids=['resident', 'generic'])
@pytest.mark.parametrize('n,W,H', [(21, 12, 200), (5, 7, 40), (33, 9, 128), (6, 5, 150), (9, 4, 224), (7, 70, 40),
                                   (3, 130, 200)])   # W > 64: the step-mask words are reloaded per 64-step chunk
def test_gru_against_oracle(n, W, H, generic, monkeypatch):
    """Both GRU kernel families (U resident on chip for H <= 200 / the general streaming kernels)."""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    from oracle import lstur_oracle as O
    if generic:
        monkeypatch.setenv('LSTUR_GRU_GENERIC', '1')
    else:
        monkeypatch.delenv('LSTUR_GRU_GENERIC', raising=False)
    rng = np.random.default_rng(2)
    U = H
    p = dict(gru_w=rng.normal(0, 0.1, (U, 3 * H)), gru_u=O.orthogonal(rng, H, 3 * H), gru_b=rng.normal(0, 0.1, 3 * H))
    x = rng.normal(0, 0.5, (n, W, U))
    gmask = rng.random((n, W)) > 0.3
    gmask[0] = False
    h0 = rng.normal(0, 0.05, (n, H))
    h_ref, cache = O.gru_fwd(p, x, gmask, h0)
    xw = (x.reshape(n * W, U) @ p['gru_w'] + p['gru_b']).astype(np.float32)
    Ud = dev(p['gru_u'].astype(np.float32))
    packed = torch.zeros(lib.lstur_gru_pack_floats(H), device='cuda')
    check(lib.lstur_gru_pack_u(Ud.data_ptr(), H, packed.data_ptr(), stream()))
    tape = torch.zeros(lib.lstur_gru_tape_floats(n, W, H), device='cuda')
    hlast = torch.empty(n, H, device='cuda')
    gm = dev(gmask.astype(np.uint8))
    xw_d, h0_d = dev(xw), dev(h0.astype(np.float32))
    check(lib.lstur_gru_fwd(xw_d.data_ptr(), gm.data_ptr(), h0_d.data_ptr(), packed.data_ptr(),
                            tape.data_ptr(), hlast.data_ptr(), n, W, H, stream()))
    np.testing.assert_allclose(hlast.cpu().numpy(), h_ref, rtol=1e-3, atol=2e-5)
    assert np.array_equal(hlast[0].cpu().numpy(), h0[0].astype(np.float32))   # fully masked row carries h0 bit-exactly

    dh = rng.normal(0, 1.0, (n, H))
    g_ref, dx_ref, dh0_ref = O.gru_bwd(p, cache, dh)
    dg = torch.empty(n, W, 3 * H, device='cuda')
    rh, hp = torch.empty(n, W, H, device='cuda'), torch.empty(n, W, H, device='cuda')
    dh0 = torch.empty(n, H, device='cuda')
    dh_d = dev(dh.astype(np.float32))
    check(lib.lstur_gru_bwd(dh_d.data_ptr(), gm.data_ptr(), tape.data_ptr(), packed.data_ptr(),
                            dg.data_ptr(), rh.data_ptr(), hp.data_ptr(), dh0.data_ptr(), n, W, H, stream()))
    np.testing.assert_allclose(hp.cpu().numpy(), cache['hs'][:, :W], rtol=1e-3, atol=2e-5)     # h_{t-1} tape
    np.testing.assert_allclose(dh0.cpu().numpy(), dh0_ref, rtol=2e-3, atol=2e-5)
    dgn = dg.cpu().numpy().reshape(n * W, 3 * H).astype(np.float64)
    np.testing.assert_allclose(dgn.sum(0), g_ref['gru_b'], rtol=2e-3, atol=1e-4)
    dU = np.zeros((H, 3 * H))
    dU[:, :2 * H] = hp.cpu().numpy().reshape(n * W, H).astype(np.float64).T @ dgn[:, :2 * H]
    dU[:, 2 * H:] = rh.cpu().numpy().reshape(n * W, H).astype(np.float64).T @ dgn[:, 2 * H:]
    np.testing.assert_allclose(dU, g_ref['gru_u'], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose((dgn @ p['gru_w'].T).reshape(n, W, U), dx_ref, rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize('arch', ['igru', 'gru', 'hgru'])
def test_full_step_at_baseline_shapes_small_batch(arch):
    """BASELINE.json configs[1] / [2] at their shapes D=300/T=30/F=400/U=200 with a batch small enough for the NumPy
    oracle: LSTUR-ini (`igru`) and the two LSTUR-con candidates, `gru` (Dense over [GRU | user], paper.py:596-599) and
    `hgru` (GRU(U/2) | Embedding(U/2), paper.py:603-607 -- the GRU kernels at H = 100)."""
    _need_gpu()
    from test_gpu_model import build, rel_close
    from conftest import make_batch
    from mnexp_amd._lib import lib, check
    dims = dict(V=2000, D=300, F=400, k=3, U=200, N_u=50)
    B, W, T, K1 = 6, 10, 30, 5
    O, p, model, rng = build(arch, dims, B, W, T, K1, 0.2)
    assert model.sizes.gru_dim == (100 if arch == 'hgru' else 200)
    user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
    ref = O.forward_backward(p, arch, user, hist, cands, drop_p=0.2, seed=77, step=0)
    u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 0, 1.0,
                                     model.stats.data_ptr(), model._stream()))
    torch.cuda.synchronize()
    logits = model.peek('logits').cpu().numpy().reshape(B, K1)
    np.testing.assert_allclose(logits, ref['logits'], rtol=1e-3, atol=1e-5)
    g = model.get_grads()
    for n in O.dense_param_names(arch):
        rel_close(g[n], ref['grads'][n], 2e-3, n)
    rel_close(g['user_row_grads'], ref['user_row_grads'], 2e-3, 'user_row_grads')


def test_fused_prefetched_steps_at_baseline_shapes_track_oracle():
    """The product path as bench.py / fit_generator drive it, at D=300/T=30/F=400/U=200: doc-id batches through the resident
    title table, every batch announced one step ahead (staging + keep bits prefetched under the recurrence), forward +
    backward + Adam in ONE call (table sweep in two halves, weight images written by the Adam launch).  Three steps
    against the oracle's train_step: per-step loss, final parameters, the touched and untouched rows of the user table."""
    _need_gpu()
    from test_gpu_model import build, rel_close
    from mnexp_amd._lib import lib
    dims = dict(V=2000, D=300, F=400, k=3, U=200, N_u=300)
    B, W, T, K1, n_docs = 6, 10, 30, 5, 120
    O, p, model, rng = build('igru', dims, B, W, T, K1, 0.2)
    tt = rng.integers(1, dims['V'], size=(n_docs, T)).astype(np.int32)
    tt[0] = 0
    for d in range(1, n_docs):
        tt[d, rng.integers(8, T + 1):] = 0
    model.set_title_table(tt)
    pf = {n: a.astype(np.float32) for n, a in p.items()}
    opt = O.Adam(pf, O.dense_param_names('igru') + ['user_emb'], lr=1e-3)
    batches = []
    for _ in range(3):
        user = rng.integers(0, dims['N_u'], size=B).astype(np.int32)
        user[2] = user[0]
        hist = rng.integers(0, n_docs, size=(B, W)).astype(np.int32)
        hist[1, :4] = 0                                   # short history: padded slots in front
        batches.append((user, hist, rng.integers(1, n_docs, size=(B, K1)).astype(np.int32)))
    dev = [tuple(model._dev_i32(a) for a in b) for b in batches]
    for i, (user, hist, cands) in enumerate(batches):
        ref = O.train_step(pf, 'igru', opt, user, tt[hist], tt[cands], drop_p=0.2, seed=77, step=i)
        if i + 1 < len(dev):
            model.prefetch_ids(dev[i + 1][1], dev[i + 1][2])
        model._train_step(*dev[i], True)
        torch.cuda.synchronize()
        assert abs(model.stats[0].item() - ref['loss']) <= 2e-3 * max(1.0, abs(ref['loss'])), i
    assert lib.lstur_prefetch_hits(model._ctx) == 2
    w = model.get_weights()
    # (1e-2 of scale: with 6 samples many conv_w / word-side gradients are ~1e-7, where Adam's m / (sqrt(v) + eps) turns a
    #  3e-8 gradient difference into a quarter of a learning-rate step)
    for n in O.dense_param_names('igru') + ['user_emb']:
        rel_close(w[n].reshape(-1), pf[n].reshape(-1), 1e-2, n)


def test_config4_shapes_batch_512_one_million_users():
    """BASELINE.json configs[3] per GPU: B = 512, 1 M users (D=300/T=30/W=50/F=400/U=200).  Logits of the full batch
    against a forward-only oracle call (fp32), then two Adam steps over the 1 M-row table: rows touched by a step against
    Keras-Adam restated on the row gradients the library reports, and the m / v DECAY of rows touched by the first step
    only (TF-1 densifies the IndexedSlices gradient: every row's moments decay every step, paper.py:657)."""
    _need_gpu()
    from oracle import lstur_oracle as O
    from mnexp_amd.native_model import NativeModel
    from mnexp_amd._lib import lib, check
    from conftest import make_batch
    dims = dict(V=3000, D=300, F=400, k=3, U=200, N_u=1000000)
    B, W, T, K1 = 512, 50, 30, 5
    rng = np.random.default_rng(3)
    p = O.init_params(rng, arch='igru', dtype=np.float32, **dims)
    model = NativeModel(p['word_emb'], dims['N_u'], B, W, T, K1 - 1, dims['F'], dims['k'], dims['U'], arch='igru',
                        dropout=0.0, lr=1e-3, seed=77)
    model.set_weights({n: p[n] for n in p if n != 'word_emb'})
    user, hist, cands = make_batch(rng, B, W, T, K1, dims['V'], dims['N_u'])
    user[:4] = [0, 999999, 123456, 999999]               # first / last row, a duplicate
    ref = O.forward_backward(p, 'igru', user, hist, cands, drop_p=0.0, want_grads=False)
    u, h, c = model._dev_i32(user), model._dev_i32(hist), model._dev_i32(cands)
    check(lib.lstur_forward_backward(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, 0, 1.0,
                                     model.stats.data_ptr(), model._stream()))
    torch.cuda.synchronize()
    logits = model.peek('logits').cpu().numpy().reshape(B, K1)
    np.testing.assert_allclose(logits, ref['logits'], rtol=1e-3, atol=2e-5)
    assert abs(model.stats[0].item() - ref['loss']) < 1e-4
    # ---- Adam step 1 on the table
    Ue, lr, b1, b2, eps = 200, 1e-3, 0.9, 0.999, 1e-7
    g = model.get_grads()
    rows1, rg1 = g['user_rows'], g['user_row_grads'].astype(np.float64)
    assert np.array_equal(rows1, user)
    table0 = model.user_emb.view(-1, Ue)[torch.from_numpy(np.unique(rows1)).cuda().long()].cpu().numpy().astype(np.float64)
    assert lib.lstur_lazy_adam(model._ctx) == 1          # a table of this size updates lazily: flush before reading it
    check(lib.lstur_adam_step(model._ctx, lr, 1, None, None, 0, model._stream()))
    model.flush_tables()
    torch.cuda.synchronize()
    uniq = np.unique(rows1)
    gsum = np.zeros((len(uniq), Ue))
    for r, gr in zip(rows1, rg1):
        gsum[np.searchsorted(uniq, r)] += gr                        # duplicates summed in ascending sample order
    m1, v1 = (1 - b1) * gsum, (1 - b2) * gsum ** 2
    lr1 = lr * np.sqrt(1 - b2) / (1 - b1)
    want1 = table0 - lr1 * m1 / (np.sqrt(v1) + eps)
    idx = torch.from_numpy(uniq).cuda().long()
    got1 = model.user_emb.view(-1, Ue)[idx].cpu().numpy()
    np.testing.assert_allclose(got1, want1, rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(model.user_m.view(-1, Ue)[idx].cpu().numpy(), m1, rtol=2e-5, atol=1e-12)
    np.testing.assert_allclose(model.user_v.view(-1, Ue)[idx].cpu().numpy(), v1, rtol=2e-5, atol=1e-20)
    probe = torch.tensor([1, 2, 500000, 999998], device='cuda')      # never touched: exactly unchanged, moments zero
    assert not model.user_m.view(-1, Ue)[probe].any() and not model.user_v.view(-1, Ue)[probe].any()
    assert torch.equal(model.user_emb.view(-1, Ue)[probe].cpu(), torch.from_numpy(p['user_emb'][probe.cpu().numpy()]))
    # ---- step 2 with other users: rows of step 1 only decay
    user2 = ((user.astype(np.int64) * 7 + 13) % dims['N_u']).astype(np.int32)
    old_only = np.setdiff1d(uniq, np.unique(user2))
    assert len(old_only) > 100
    u2 = model._dev_i32(user2)
    check(lib.lstur_forward_backward(model._ctx, u2.data_ptr(), h.data_ptr(), c.data_ptr(), B, 1, 1.0,
                                     model.stats.data_ptr(), model._stream()))
    check(lib.lstur_adam_step(model._ctx, lr, 2, None, None, 0, model._stream()))
    torch.cuda.synchronize()
    # lazily, the rows of step 1 that step 2 did not touch are still at iteration 1 ...
    io_ = torch.from_numpy(old_only).cuda().long()
    stale = model.user_m.view(-1, Ue)[io_].cpu().numpy()
    np.testing.assert_allclose(stale, m1[np.searchsorted(uniq, old_only)], rtol=2e-5, atol=1e-12)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.flush_tables()                                 # ... and take their zero-gradient update of iteration 2 here
    e1.record()
    torch.cuda.synchronize()
    # (what the periodic full flush of the lr_t ring costs: once per 65 534 iterations at the product ring length)
    print('full flush of the lazily updated table at 1 M users x 200 (one pending iteration): %.2f ms' % e0.elapsed_time(e1))
    sel = np.searchsorted(uniq, old_only)
    io = torch.from_numpy(old_only).cuda().long()
    m2, v2 = b1 * m1[sel], b2 * v1[sel]
    lr2 = lr * np.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
    want2 = got1[sel].astype(np.float64) - lr2 * m2 / (np.sqrt(v2) + eps)
    np.testing.assert_allclose(model.user_m.view(-1, Ue)[io].cpu().numpy(), m2, rtol=2e-5, atol=1e-12)
    np.testing.assert_allclose(model.user_v.view(-1, Ue)[io].cpu().numpy(), v2, rtol=2e-5, atol=1e-20)
    np.testing.assert_allclose(model.user_emb.view(-1, Ue)[io].cpu().numpy(), want2, rtol=2e-5, atol=2e-7)


def _split3_np(x):
    """bf16_mfma.h: split3 -- three round-to-nearest-even bf16 parts of an fp32 array (as uint16 bit patterns)."""
    def rne(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
        return r.astype(np.uint16)
    def up(b):
        return (b.astype(np.uint32) << 16).view(np.float32)
    h = rne(x)
    r1 = (x - up(h)).astype(np.float32)
    m = rne(r1)
    r2 = (r1 - up(m)).astype(np.float32)
    return h, m, rne(r2)


def _pow2_scale_np(mx, target):
    """bf16_mfma.h: pow2_scale -- the power of two s with mx * s in [2^target, 2^(target + 1)) (1 for mx == 0)."""
    mx = np.float32(mx)
    if mx == 0:
        return np.float32(1.0)
    return np.float32(2.0 ** (target - int(np.floor(np.log2(float(mx))))))


def _split2_np(x, s):
    """bf16_mfma.h: split2 of x * s -- h = fp16(x s), l = fp16(x s - h), round-to-nearest-even (uint16 bit patterns)."""
    v = (x.astype(np.float32) * np.float32(s)).astype(np.float32)
    h = v.astype(np.float16)
    l = (v - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h.view(np.uint16), l.view(np.uint16)


def _convx_scales(lib, check, table_d=None, count=0, drop=0.0):
    """a fresh scale block of the f16x2 kernels (+ the embedding scale of `table_d`)"""
    cs = torch.zeros(lib.lstur_convx_scales_bytes() // 4, dtype=torch.float32, device='cuda')
    check(lib.lstur_convx_scales_init(cs.data_ptr(), stream()))
    if table_d is not None:
        check(lib.lstur_convx_embedding_scale(table_d.data_ptr(), count, drop, cs.data_ptr(), stream()))
    return cs


@pytest.mark.parametrize('parts', [2, 3])
@pytest.mark.parametrize('N,D,T,k,drop', [(37, 300, 30, 3, 0.2), (5, 12, 7, 3, 0.0), (6, 20, 6, 5, 0.5), (3, 48, 11, 1, 0.2)])
def test_convx_gather_is_the_split_of_the_fp32_gather(N, D, T, k, drop, parts):
    """convx.hip: gatherx_kernel writes the 16-bit parts of exactly the values lstur_gather_f32 writes (same dropout mask,
    same 1 / (1 - p)) in the Ex layout [title][Tp][Dp/8][part][8] with zero pad elements -- bit for bit: parts = 3 the bf16
    triple h | m | l, parts = 2 the fp16 pair h | l of the value times the power of two that puts max |table| / (1 - p) into
    [2^14, 2^15) (restated here in NumPy)."""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    rng = np.random.default_rng(D + N)
    V = 60
    table = rng.normal(0, 0.3, size=(V, D)).astype(np.float32)
    ids = rng.integers(0, V, size=(N, T)).astype(np.int32)
    ids[:, max(1, T - 3):] = 0
    pl = (k - 1) // 2
    pr = k - 1 - pl
    Tp, Dp = T + k - 1, (D + 7) // 8 * 8
    t_d, i_d = dev(table), dev(ids)
    ref = torch.full((N * Tp * D + 256,), 9.0, dtype=torch.float32, device='cuda')
    check(lib.lstur_gather_f32(t_d.data_ptr(), i_d.data_ptr(), ref.data_ptr(), N, T, D, pl, pr, drop, 1234, stream()))
    cs = _convx_scales(lib, check, t_d, V * D, drop) if parts == 2 else None
    csp = cs.data_ptr() if parts == 2 else None
    ne = lib.lstur_convx_e_elems(N, T, D, k, parts)
    assert ne >= N * Tp * Dp * parts
    ex = torch.full((ne,), 0x7FC0, dtype=torch.int16, device='cuda')
    check(lib.lstur_convx_gather(t_d.data_ptr(), i_d.data_ptr(), ex.data_ptr(), N, T, D, pl, pr, drop, 1234, 0, parts, csp, None, stream()))
    torch.cuda.synchronize()
    E = ref[:N * Tp * D].cpu().numpy().reshape(N, Tp, D)
    Epad = np.zeros((N, Tp, Dp), np.float32)
    Epad[:, :, :D] = E
    if parts == 3:
        pieces = _split3_np(Epad)
    else:
        ds = np.float32(1.0) / (np.float32(1.0) - np.float32(drop)) if drop > 0 else np.float32(1.0)
        sE = _pow2_scale_np(np.abs(table).max() * ds, 14)
        assert float(cs[0].item()) == float(sE)
        assert np.abs(Epad * sE).max() < 65504
        pieces = _split2_np(Epad, sE)
    want = np.stack(pieces, axis=0).reshape(parts, N, Tp, Dp // 8, 8).transpose(1, 2, 3, 0, 4)    # [N][Tp][C8][part][8]
    got = ex[:N * Tp * Dp * parts].cpu().numpy().view(np.uint16).reshape(N, Tp, Dp // 8, parts, 8)
    assert np.array_equal(got, want)
    # skip_unread: exactly the rows some non-pad token's window covers are written (same values), the others untouched
    nonpad = np.zeros((N, Tp + k), bool)
    for j in range(k):                                    # token t covers padded rows t .. t + k - 1
        nonpad[:, j:j + T] |= ids != 0
    need = nonpad[:, :Tp]
    # ... found from the neighbours' ids, or from the titles' non-pad bit masks (what the training step hands over)
    live = dev(((ids != 0).astype(np.uint64) << np.arange(T, dtype=np.uint64)).sum(1).astype(np.uint32).view(np.int32))
    for lv in (None, live.data_ptr()):
        exs = torch.full((ne,), 0x1234, dtype=torch.int16, device='cuda')
        check(lib.lstur_convx_gather(t_d.data_ptr(), i_d.data_ptr(), exs.data_ptr(), N, T, D, pl, pr, drop, 1234, 1, parts, csp, lv,
                                     stream()))
        gots = exs[:N * Tp * Dp * parts].cpu().numpy().view(np.uint16).reshape(N, Tp, Dp // 8, parts, 8)
        assert np.array_equal(gots[need], want[need])
        assert (gots[~need] == 0x1234).all() and (~need).any()


def _convx_case(lib, check, rng, N, D, T, F, k, parts, sliced, table, Wc, dC, tag, bias_scale=1.0):
    """forward + kernel gradient of the split-operand conv family on one case: errors against fp64 and against the
    exact-fp32 MFMA kernels; returns them as fractions of the result scales"""
    V = table.shape[0]
    Tp, pl = T + k - 1, (k - 1) // 2
    ids = rng.integers(0, V, size=(N, T)).astype(np.int32)
    ids[:, max(1, T - 4):] = 0
    ids[0, 1] = 0                                           # a pad token INSIDE a title: its row 0 embedding is gathered
    b = (rng.normal(0, 0.1, size=F) * bias_scale).astype(np.float32)
    t_d, ids_d, b_d = dev(table), dev(ids), dev(b)
    cs = _convx_scales(lib, check, t_d, V * D, 0.0) if parts == 2 else None
    csp = cs.data_ptr() if parts == 2 else None
    ex = torch.zeros(lib.lstur_convx_e_elems(N, T, D, k, parts), dtype=torch.int16, device='cuda')
    check(lib.lstur_convx_gather(t_d.data_ptr(), ids_d.data_ptr(), ex.data_ptr(), N, T, D, pl, k - 1 - pl, 0.0, 0, 0, parts, csp, None, stream()))
    Wd = dev(Wc.reshape(k * D, F))
    wx = torch.zeros(lib.lstur_convx_w_elems(D, k, F, parts), dtype=torch.int16, device='cuda')
    check(lib.lstur_convx_pack_w(Wd.data_ptr(), D, k, F, wx.data_ptr(), parts, csp, stream()))
    out = torch.full((N * T, F), -7.0, dtype=torch.float32, device='cuda')
    scr = torch.empty(lib.lstur_convx_fwd_scratch(), dtype=torch.uint8, device='cuda') if sliced else None
    scr_p = scr.data_ptr() if sliced else None
    check(lib.lstur_convx_fwd(ex.data_ptr(), wx.data_ptr(), b_d.data_ptr(), ids_d.data_ptr(), out.data_ptr(), N, T, D, k, F,
                              None, None, parts, csp, scr_p, stream()))
    E = dev(table)[dev(ids).long()]                        # [N, T, D]
    x = torch.nn.functional.pad(E.double().transpose(1, 2), (pl, k - 1 - pl))
    ref = torch.relu(torch.nn.functional.conv1d(x, dev(Wc).double().permute(2, 1, 0), dev(b).double()))
    ref = ref.transpose(1, 2) * (dev(ids) != 0).unsqueeze(-1)
    scale = max(1.0, ref.abs().max().item()) if bias_scale else ref.abs().max().item()
    r = {}
    assert torch.isfinite(out).all()
    r['fwd_vs_f64'] = (out.view(N, T, F).double() - ref).abs().max().item() / scale
    assert (out.view(N, T, F)[dev(ids) == 0] == 0).all()
    # the exact-fp32 MFMA kernel on the fp32 gather
    Epd = torch.zeros(N * Tp * D + 256, dtype=torch.float32, device='cuda')
    check(lib.lstur_gather_f32(t_d.data_ptr(), ids_d.data_ptr(), Epd.data_ptr(), N, T, D, pl, k - 1 - pl, 0.0, 0, stream()))
    Wp = pack_b(Wd)
    out32 = torch.empty(N * T, F, dtype=torch.float32, device='cuda')
    check(lib.lstur_conv1d_relu_mask_fwd(Epd.data_ptr(), Wp.data_ptr(), b_d.data_ptr(), ids_d.data_ptr(),
                                         out32.data_ptr(), N, T, D, k, F, None, None, stream()))
    r['f32_fwd_vs_f64'] = (out32.view(N, T, F).double() - ref).abs().max().item() / scale
    r['fwd_vs_f32'] = (out - out32).abs().max().item() / scale
    # over the compacted list of non-pad tokens: same rows bit-identical, pad rows untouched
    vidx = torch.full((N * T,), -1, dtype=torch.int32, device='cuda')
    vcount = torch.zeros(1, dtype=torch.int32, device='cuda')
    vs = torch.zeros(max(1, lib.lstur_valid_tokens_scratch(N * T) // 4), dtype=torch.int32, device='cuda')
    check(lib.lstur_valid_tokens(ids_d.data_ptr(), N * T, vidx.data_ptr(), vcount.data_ptr(), vs.data_ptr(), stream()))
    out2 = torch.full((N * T, F), -7.0, dtype=torch.float32, device='cuda')
    check(lib.lstur_convx_fwd(ex.data_ptr(), wx.data_ptr(), b_d.data_ptr(), ids_d.data_ptr(), out2.data_ptr(), N, T, D, k, F,
                              vidx.data_ptr(), vcount.data_ptr(), parts, csp, scr_p, stream()))
    valid = dev(ids.reshape(-1) != 0)
    if sliced:     # a different token count can cut the tail tiles into a different number of k slices
        assert (out2[valid] - out[valid]).abs().max().item() < 2e-6 * scale, tag
    else:
        assert torch.equal(out2[valid], out[valid]), tag
    assert (out2[~valid] == -7.0).all(), tag
    # ---- kernel gradient on the same Ex: against fp64, and over the compacted list with NaN in the rows it must not read
    dC_d = dev(dC)
    Ep_d = Epd[:N * Tp * D].view(N, Tp, D)
    cols = torch.stack([Ep_d[:, j:j + T] for j in range(k)], dim=2).reshape(N * T, k * D).double()
    ref_dw = cols.t() @ dC_d.double()
    dw = torch.full((k * D, F), 3.0, dtype=torch.float32, device='cuda')
    wscr = torch.empty(lib.lstur_convx_wgrad_scratch(N, T, D, k, F), dtype=torch.uint8, device='cuda')
    offs = torch.zeros(2 * N * T, dtype=torch.int32, device='cuda')
    check(lib.lstur_convx_wgrad(ex.data_ptr(), dC_d.data_ptr(), dw.data_ptr(), wscr.data_ptr(), offs.data_ptr(), N, T, D, k, F,
                                None, None, parts, csp, stream()))
    wscale = ref_dw.abs().max().item()
    assert torch.isfinite(dw).all()
    r['wgrad_vs_f64'] = (dw.double() - ref_dw).abs().max().item() / wscale
    dw32 = torch.full((k * D, F), 3.0, dtype=torch.float32, device='cuda')
    s32 = torch.empty(lib.lstur_conv1d_wgrad_scratch(D, k, F), dtype=torch.uint8, device='cuda')
    check(lib.lstur_conv1d_wgrad(Epd.data_ptr(), dC_d.data_ptr(), dw32.data_ptr(), s32.data_ptr(), N, T, D, k, F, None, None, stream()))
    r['f32_wgrad_vs_f64'] = (dw32.double() - ref_dw).abs().max().item() / wscale
    ref2 = cols.t() @ (dC_d * valid.unsqueeze(-1)).double()
    dC_g = torch.where(valid.unsqueeze(-1), dC_d, torch.full_like(dC_d, float('nan')))   # pad rows must not be read
    dw2 = torch.full((k * D, F), 3.0, dtype=torch.float32, device='cuda')
    check(lib.lstur_convx_wgrad(ex.data_ptr(), dC_g.data_ptr(), dw2.data_ptr(), wscr.data_ptr(), offs.data_ptr(), N, T, D, k, F,
                                vidx.data_ptr(), vcount.data_ptr(), parts, csp, stream()))
    r['wgrad_list_vs_f64'] = (dw2.double() - ref2).abs().max().item() / max(ref2.abs().max().item(), 1e-30)
    return r


@pytest.mark.parametrize('parts', [2, 3])
@pytest.mark.parametrize('sliced', [True, False])
@pytest.mark.parametrize('N,D,T,F,k', [(130, 300, 30, 400, 3), (3, 300, 30, 400, 3), (9, 12, 7, 16, 3), (7, 20, 6, 24, 5),
                                       (11, 48, 11, 448, 1), (5, 16, 8, 230, 3), (40, 300, 30, 250, 3),
                                       (1200, 300, 30, 400, 3)])
def test_convx_presplit_is_fp32_grade(N, D, T, F, k, sliced, parts):
    """convx.hip: Conv1D forward and kernel gradient on PRE-SPLIT operands (Ex from the gather, k'-ordered split weights),
    parts = 2 (f16x2: three fp16 products on power-of-two scaled data) and 3 (bf16x3: six bf16 products).  Both are held to
    the SAME bounds as the exact-fp32 MFMA kernels: 2e-5 of the result scale against fp64, 4e-6 against the exact-fp32
    kernel, and at most twice that kernel's own error against fp64 (+ 3e-7 of scale: the floor of a 2^-22 operand split).
    `sliced`: with the scratch for the k-sliced tail tiles (tiles past the last whole round of 256 are cut into <= 8 k slices
    summed by the fix-up kernel; N = 1200 gives 282 tiles = one round + a sliced tail)."""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    rng = np.random.default_rng(N * 7 + F)
    table = rng.normal(0, 0.3, size=(70, D)).astype(np.float32)
    Wc = rng.normal(0, 0.05, size=(k, D, F)).astype(np.float32)
    dC = rng.normal(size=(N * T, F)).astype(np.float32)
    r = _convx_case(lib, check, rng, N, D, T, F, k, parts, sliced, table, Wc, dC, 'plain')
    print('convx parts=%d N=%d D=%d F=%d k=%d: %s' % (parts, N, D, F, k, {a: '%.2e' % b for a, b in r.items()}))
    assert r['fwd_vs_f64'] < 2e-5 and r['fwd_vs_f32'] < 4e-6, r
    assert r['fwd_vs_f64'] <= 2 * r['f32_fwd_vs_f64'] + 3e-7, r
    assert r['wgrad_vs_f64'] < 2e-5 and r['wgrad_list_vs_f64'] < 2e-5, r
    assert r['wgrad_vs_f64'] <= 2 * r['f32_wgrad_vs_f64'] + 3e-7, r


@pytest.mark.parametrize('case', ['outliers', 'tiny', 'huge', 'mixed_gradient'])
def test_convx_f16x2_adversarial_dynamic_range(case):
    """The fp16 pair covers 11 + 11 bits below the tensor's LARGEST magnitude, with an absolute floor 2^-37 under it: operands
    whose magnitudes spread over many binades (outliers in the word table / the weight / the gradient), tensors far outside
    the fp16 range (1e-12, 3e6) -- the power-of-two scales must bring every one of them in, no overflow, no flush that shows
    at the result's scale.  Bounds as in the plain test."""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    N, D, T, F, k = 40, 300, 30, 400, 3
    rng = np.random.default_rng(11)
    table = rng.normal(0, 0.3, size=(70, D)).astype(np.float32)
    Wc = rng.normal(0, 0.05, size=(k, D, F)).astype(np.float32)
    dC = rng.normal(size=(N * T, F)).astype(np.float32)
    if case == 'outliers':          # a few elements 2^10 .. 2^14 times the bulk
        table[rng.integers(0, 70, 20), rng.integers(0, D, 20)] *= 3000.0
        Wc.reshape(-1)[rng.integers(0, Wc.size, 50)] *= 1000.0
        dC.reshape(-1)[rng.integers(0, dC.size, 50)] *= 16000.0
    elif case == 'tiny':
        table *= 1e-12
        Wc *= 1e-9
        dC *= 1e-20
    elif case == 'huge':
        table *= 3e6
        Wc *= 1e5
        dC *= 1e9
    else:                           # gradient rows whose scales differ by 2^30 (early / late training samples in one batch)
        dC *= np.exp2(rng.integers(-30, 1, size=(N * T, 1))).astype(np.float32)
    r = _convx_case(lib, check, rng, N, D, T, F, k, 2, True, table, Wc, dC, case, bias_scale=0.0)
    print('convx f16x2 %s: %s' % (case, {a: '%.2e' % b for a, b in r.items()}))
    assert r['fwd_vs_f64'] < 2e-5 and r['fwd_vs_f64'] <= 2 * r['f32_fwd_vs_f64'] + 3e-7, r
    assert r['wgrad_vs_f64'] < 2e-5 and r['wgrad_vs_f64'] <= 2 * r['f32_wgrad_vs_f64'] + 3e-7, r
    assert r['wgrad_list_vs_f64'] < 2e-5, r


@pytest.mark.parametrize('N,T,F,drop', [(37, 30, 400, 0.2), (9, 32, 512, 0.0), (21, 11, 80, 0.25), (5, 30, 404, 0.2)])
def test_attpool_title_kernels_against_fp64(N, T, F, drop):
    """Masking -> Dropout -> SimpleAttentionMaskSupport (paper.py:157-158, models.py:474-489) forward and backward through
    lstur_attpool_fwd / _bwd (the title-resident kernels) against the formulas in fp64.  Pad tokens sit at the tail, in the
    middle and over whole titles; their conv rows hold NaN (the kernels must neither read nor write them); a live token whose
    conv row is all zero is masked out of the softmax (Masking = any(c != 0))."""
    _need_gpu()
    from mnexp_amd._lib import lib, check
    from oracle import lstur_oracle as O
    rng = np.random.default_rng(3)
    ids = rng.integers(1, 50, size=(N, T)).astype(np.int32)
    for n in range(N):
        ids[n, rng.integers(T // 2, T + 1):] = 0          # padded tail (possibly none)
    ids[1, 2] = 0                                          # a pad inside a title
    ids[2] = 0                                             # an all-pad title
    c = np.maximum(rng.normal(0, 1, size=(N, T, F)), 0).astype(np.float32)
    c[3, 1] = 0                                            # live token, all-zero conv row: Masking drops it
    live = ids != 0
    k = rng.normal(0, 0.2, size=F).astype(np.float32)
    b = np.float32(0.1)
    dp = rng.normal(0, 1, size=(N, F)).astype(np.float32)
    key = O.dropout_key(7, 3, 1)
    keep = np.ones(N * T * F, bool) if drop == 0 else O.dropout_keep(key, N * T * F, drop)
    scale = 1.0 / (1.0 - drop)
    x = np.where(live[..., None], c.astype(np.float64), 0.0) * keep.reshape(N, T, F) * scale
    a = np.where(live, np.tanh(x @ k.astype(np.float64) + np.float64(b)), 0.0)
    m = live & (c != 0).any(-1)
    e = np.exp(a) * m
    w = e / (e.sum(-1, keepdims=True) + 1e-7)
    pooled = (x * w[..., None]).sum(1)
    dw = (x * dp[:, None, :].astype(np.float64)).sum(-1)
    ds = (dw - (dw * w).sum(-1, keepdims=True)) * w * (1 - a * a)
    dx = dp[:, None, :].astype(np.float64) * w[..., None] + ds[..., None] * k.astype(np.float64)
    dc = dx * keep.reshape(N, T, F) * scale * (c > 0) * live[..., None]
    c_dev = c.copy()
    c_dev[~live] = np.nan
    cd, idd, kd, bd, dpd = dev(c_dev), dev(ids), dev(k), dev(np.array([b], np.float32)), dev(dp)
    f32 = dict(dtype=torch.float32, device='cuda')
    pool_g, a_g, w_g = torch.zeros(N, F, **f32), torch.zeros(N, T, **f32), torch.zeros(N, T, **f32)
    check(lib.lstur_attpool_fwd(cd.data_ptr(), idd.data_ptr(), kd.data_ptr(), bd.data_ptr(), pool_g.data_ptr(), a_g.data_ptr(),
                                w_g.data_ptr(), N, T, F, drop, key, stream()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(a_g.cpu().numpy(), a, rtol=0, atol=2e-6)
    np.testing.assert_allclose(w_g.cpu().numpy(), w, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(pool_g.cpu().numpy(), pooled, rtol=2e-5, atol=2e-6)
    nparts = lib.lstur_attpool_bwd_parts(N)
    dc_g = torch.full((N, T, F), float('nan'), **f32)
    parts = torch.zeros(nparts, 2 * F + 1, **f32)
    scratch = torch.zeros(N * T, **f32)
    check(lib.lstur_attpool_bwd(cd.data_ptr(), idd.data_ptr(), kd.data_ptr(), a_g.data_ptr(), w_g.data_ptr(), dpd.data_ptr(),
                                dc_g.data_ptr(), parts.data_ptr(), scratch.data_ptr(), N, T, F, drop, key, stream()))
    torch.cuda.synchronize()
    got = dc_g.cpu().numpy()
    assert np.isnan(got[~live]).all()                      # pad rows untouched
    np.testing.assert_allclose(got[live], dc[live], rtol=2e-5, atol=2e-6)
    red = parts.cpu().numpy().astype(np.float64).sum(0)
    np.testing.assert_allclose(red[:F], (ds[..., None] * x).sum((0, 1)), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(red[F:2 * F], dc.sum((0, 1)), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(red[2 * F], ds.sum(), rtol=1e-4, atol=1e-5)
