"""Pins the oracle's hand-written backward against torch autograd (fp64) and
the dropout-hash / Adam restatements against direct formulas."""
import numpy as np
import pytest
import torch

from conftest import make_batch
from oracle import lstur_oracle as O
from oracle import torch_ref as TR

DIMS = dict(V=60, D=12, F=16, k=3, U=8, N_u=11)


@pytest.mark.parametrize("arch", ["igru", "gru", "hgru", "nigru"])
@pytest.mark.parametrize("drop", [0.0, 0.25])
def test_grads_match_autograd(arch, drop):
    rng = np.random.default_rng(3)
    p = O.init_params(rng, arch=arch, dtype=np.float64, **DIMS)
    for n in ("conv_b", "att_b", "dense_b", "gru_b"):
        p[n] = rng.normal(0, 0.1, size=p[n].shape)
    B, W, T, K1 = 5, 4, 7, 3
    user, hist, cands = make_batch(rng, B, W, T, K1, DIMS["V"], DIMS["N_u"])
    user[1] = user[0]  # duplicate user rows must sum
    out = O.forward_backward(p, arch, user, hist, cands, drop_p=drop, seed=7, step=2, word_trainable=True)

    names = [n for n in p]
    tp = TR.to_t(p, grad_names=names)
    keep1 = keep2 = None
    scale = 1.0
    if drop > 0:
        N_t = B * (W + K1)
        keep1 = O.dropout_keep(O.dropout_key(7, 2, 0), N_t * T * DIMS["D"], drop).reshape(N_t, T, DIMS["D"])
        keep2 = O.dropout_keep(O.dropout_key(7, 2, 1), N_t * T * DIMS["F"], drop).reshape(N_t, T, DIMS["F"])
        scale = 1.0 / (1.0 - drop)
        assert 0.6 < keep1.mean() < 0.9
    loss, z, pr = TR.model_loss(tp, arch, user, hist, cands, keep1, keep2, scale)
    loss.backward()
    assert abs(loss.item() - out["loss"]) < 1e-12
    np.testing.assert_allclose(out["logits"], z.detach().numpy(), rtol=1e-10, atol=1e-12)
    g = dict(out["grads"])
    g["user_emb"] = O.densify_user_grad(*p["user_emb"].shape, out["user_rows"], out["user_row_grads"], np.float64)
    for n in names:
        if tp[n].grad is None:                      # nigru: the user table is not part of the graph
            assert n == "user_emb" and not g[n].any()
            continue
        ref = tp[n].grad.numpy()
        np.testing.assert_allclose(g[n], ref, rtol=1e-8, atol=1e-12, err_msg=n)


@pytest.mark.parametrize("arch", ["igru", "gru", "hgru", "pgru", "vo", "iigru", "nigru"])
def test_id_keep_user_embedding_masking_matches_autograd(arch):
    """`id_keep` (cook.py:139-142: idx_vec * Dropout(1 - id_keep)(idx_mask)): a training sample keeps its user embedding
    with probability id_keep, kept rows scaled by 1 / id_keep; the row gradients pass the same mask.  Against autograd."""
    rng = np.random.default_rng(9)
    p = O.init_params(rng, arch=arch, dtype=np.float64, **DIMS)
    B, W, T, K1 = 12, 4, 7, 3
    user, hist, cands = make_batch(rng, B, W, T, K1, DIMS["V"], DIMS["N_u"])
    out = O.forward_backward(p, arch, user, hist, cands, drop_p=0.0, seed=7, step=2, id_keep=0.6)
    keep_u = O.dropout_keep(O.dropout_key(7, 2, 2), B, 0.4)
    assert 0 < keep_u.sum() < B
    names = [n for n in p if n != "word_emb"]
    tp = TR.to_t(p, grad_names=names)
    has_table = arch != "nigru"
    loss, z, _ = TR.model_loss(tp, arch, user, hist, cands, keep_u=keep_u if has_table else None, keep_u_scale=1.0 / 0.6)
    loss.backward()
    assert abs(loss.item() - out["loss"]) < 1e-12
    g = dict(out["grads"])
    g["user_emb"] = O.densify_user_grad(*p["user_emb"].shape, out["user_rows"], out["user_row_grads"], np.float64)
    if arch == "iigru":
        g["user_emb2"] = O.densify_user_grad(*p["user_emb2"].shape, out["user_rows"], out["user_row_grads2"], np.float64)
    for n in names:
        if tp[n].grad is None:          # not part of this arch's graph (nigru: the table; vo: the GRU)
            assert not g[n].any(), n
            continue
        np.testing.assert_allclose(g[n], tp[n].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=n)
    if has_table:
        assert not out["user_row_grads"][~keep_u].any()          # a dropped sample trains no user row


@pytest.mark.parametrize("arch", ["pgru", "vo", "niavg", "niatt", "iigru"])
def test_minor_user_encoders_match_autograd(arch):
    """`-a pgru | vo | niavg` (paper.py:622-630): GRU + user vector, the user vector alone, masked mean of the history."""
    rng = np.random.default_rng(5)
    p = O.init_params(rng, arch=arch, dtype=np.float64, **DIMS)
    for n in [n for n in p if n.endswith("_b")]:
        p[n] = rng.normal(0, 0.1, size=p[n].shape)
    B, W, T, K1 = 5, 4, 7, 3
    user, hist, cands = make_batch(rng, B, W, T, K1, DIMS["V"], DIMS["N_u"])
    user[1] = user[0]
    hist[2] = 0
    out = O.forward_backward(p, arch, user, hist, cands, drop_p=0.0)
    names = [n for n in p if n != "word_emb"]
    tp = TR.to_t(p, grad_names=names)
    loss, z, pr = TR.model_loss(tp, arch, user, hist, cands)
    loss.backward()
    assert abs(loss.item() - out["loss"]) < 1e-12
    g = dict(out["grads"])
    g["user_emb"] = O.densify_user_grad(*p["user_emb"].shape, out["user_rows"], out["user_row_grads"], np.float64)
    if arch == "iigru":
        g["user_emb2"] = O.densify_user_grad(*p["user_emb2"].shape, out["user_rows"], out["user_row_grads2"], np.float64)
    for n in names:
        ref = tp[n].grad.numpy() if tp[n].grad is not None else np.zeros_like(p[n])
        np.testing.assert_allclose(g[n], ref, rtol=1e-8, atol=1e-12, err_msg=n)


@pytest.mark.parametrize("arch,score_model", [("igru", "ddot"), ("igru", "dnn"), ("gru", "ddot"), ("ngru", "dnn"),
                                              ("ngru", "ddot"), ("dgru", "dnn"), ("dgru", "ddot"), ("hgru", "dnn")])
def test_score_models_and_wide_user_vectors_match_autograd(arch, score_model):
    """`--score-model ddot | dnn` (paper.py:448-455) and the 2U-wide user vectors of `-a ngru | dgru`
    (paper.py:600-602, 608-611; dgru drops the whole user embedding of a sample with p = 0.5)."""
    rng = np.random.default_rng(11)
    p = O.init_params(rng, arch=arch, dtype=np.float64, score_model=score_model, **DIMS)
    for n in p:
        if n.endswith("_b"):
            p[n] = rng.normal(0, 0.1, size=p[n].shape)
    B, W, T, K1 = 6, 4, 7, 3
    user, hist, cands = make_batch(rng, B, W, T, K1, DIMS["V"], DIMS["N_u"])
    user[1] = user[0]
    out = O.forward_backward(p, arch, user, hist, cands, drop_p=0.0, seed=7, step=2)
    names = [n for n in p if n != "word_emb"]
    tp = TR.to_t(p, grad_names=names)
    keep_u = O.dropout_keep(O.dropout_key(7, 2, 2), B, 0.5) if arch == "dgru" else None
    if arch == "dgru":
        assert 0 < keep_u.sum() < B
    loss, z, pr = TR.model_loss(tp, arch, user, hist, cands, keep_u=keep_u)
    loss.backward()
    assert abs(loss.item() - out["loss"]) < 1e-12
    np.testing.assert_allclose(out["logits"], z.detach().numpy(), rtol=1e-10, atol=1e-12)
    g = dict(out["grads"])
    g["user_emb"] = O.densify_user_grad(*p["user_emb"].shape, out["user_rows"], out["user_row_grads"], np.float64)
    assert set(O.dense_param_names(arch, score_model=score_model)) <= set(g)
    for n in names:
        np.testing.assert_allclose(g[n], tp[n].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=n)
    # test model: sigmoid(score_model([u, d])) without dropout
    pr1 = O.predict(p, arch, user, hist, cands[:, 1])
    _, z1, _ = TR.model_loss(tp, arch, user, hist, cands[:, 1:2])
    np.testing.assert_allclose(pr1, torch.sigmoid(z1[:, 0]).detach().numpy(), rtol=1e-10)


@pytest.mark.parametrize("drop", [0.0, 0.25])
def test_mhsa_grads_match_autograd(drop, monkeypatch):
    """The build-defined multi-head self-attention news encoder (no reference counterpart: SURVEY.md section 8 row
    a17): the oracle's hand-derived backward, bf16 emulation off, against torch autograd in fp64."""
    monkeypatch.setattr(O, "MHSA_EMULATE_BF16", False)
    rng = np.random.default_rng(4)
    dims = dict(V=60, D=12, F=32, k=1, U=8, N_u=11)
    p = O.init_params(rng, arch="igru", dtype=np.float64, encoder="mhsa", heads=2, **dims)
    for n in ("qkv_b", "att_b", "dense_b", "gru_b"):
        p[n] = rng.normal(0, 0.1, size=p[n].shape)
    B, W, T, K1 = 5, 4, 7, 3
    user, hist, cands = make_batch(rng, B, W, T, K1, dims["V"], dims["N_u"])
    hist[2] = 0
    out = O.forward_backward(p, "igru", user, hist, cands, drop_p=drop, seed=7, step=2, word_trainable=True)
    names = [n for n in p if n != "mhsa_heads"]
    tp = TR.to_t({n: p[n] for n in names}, grad_names=names)
    tp["mhsa_heads"] = 2
    keep1 = keep2 = None
    scale = 1.0
    if drop > 0:
        N_t = B * (W + K1)
        keep1 = O.dropout_keep(O.dropout_key(7, 2, 0), N_t * T * dims["D"], drop).reshape(N_t, T, dims["D"])
        keep2 = O.dropout_keep(O.dropout_key(7, 2, 1), N_t * T * dims["F"], drop).reshape(N_t, T, dims["F"])
        scale = 1.0 / (1.0 - drop)
    loss, z, pr = TR.model_loss(tp, "igru", user, hist, cands, keep1, keep2, scale)
    loss.backward()
    assert abs(loss.item() - out["loss"]) < 1e-12
    np.testing.assert_allclose(out["logits"], z.detach().numpy(), rtol=1e-10, atol=1e-12)
    g = dict(out["grads"])
    g["user_emb"] = O.densify_user_grad(*p["user_emb"].shape, out["user_rows"], out["user_row_grads"], np.float64)
    for n in names:
        np.testing.assert_allclose(g[n], tp[n].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=n)


def test_bf16_round_is_round_to_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.1415927, 1e-30, 0.0], dtype=np.float32)
    got = O.bf16_round(x)
    ref = torch.tensor(x).to(torch.bfloat16).to(torch.float32).numpy()
    np.testing.assert_array_equal(got, ref)
    r = np.random.default_rng(0).normal(size=4096).astype(np.float32)
    np.testing.assert_array_equal(O.bf16_round(r), torch.tensor(r).to(torch.bfloat16).to(torch.float32).numpy())


def test_all_masked_history_returns_initial_state():
    rng = np.random.default_rng(0)
    p = O.init_params(rng, arch="igru", dtype=np.float64, **DIMS)
    user, hist, cands = make_batch(rng, 3, 4, 7, 2, DIMS["V"], DIMS["N_u"])
    hist[:] = 0
    out = O.forward_backward(p, "igru", user, hist, cands, want_grads=False)
    np.testing.assert_array_equal(out["user_vec"], p["user_emb"][user])


def test_dropout_hash_scalar_restatement():
    def fmix(h):
        h &= 0xFFFFFFFF
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        h ^= h >> 16
        return h

    seed, step, stream = 1234, 17, 1
    k = fmix((seed + 0x9E3779B9 * (stream + 1)) & 0xFFFFFFFF)
    k = fmix(k ^ ((step * 0x7FEB352D + 0x165667B1) & 0xFFFFFFFF))
    assert k == O.dropout_key(seed, step, stream)
    keep = O.dropout_keep(k, 1000, 0.2, start=5)
    thr = int(0.2 * 65536.0)
    for i in range(1000):
        e = i + 5
        h = fmix(((e >> 1) * 0x9E3779B1 + k) & 0xFFFFFFFF)        # one hash per pair of elements
        assert keep[i] == (((h >> 16) if e & 1 else (h & 0xFFFF)) >= thr)
    assert abs(O.dropout_keep(k, 200000, 0.2).mean() - 0.8) < 3e-3


def test_adam_matches_keras_formula():
    rng = np.random.default_rng(1)
    p = {"w": rng.normal(size=(4, 3)).astype(np.float32)}
    opt = O.Adam(p, ["w"], lr=1e-3)
    w = p["w"].astype(np.float64).copy()
    m = np.zeros_like(w)
    v = np.zeros_like(w)
    for t in range(1, 6):
        g = rng.normal(size=(4, 3)).astype(np.float32)
        opt.step(p, {"w": g})
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g.astype(np.float64) ** 2
        w = w - lr_t * m / (np.sqrt(v) + 1e-7)
        np.testing.assert_allclose(p["w"], w, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("arch", ["igru", "nigru", "ngru", "iicgru"])
@pytest.mark.parametrize("head", ["point_mlp", "point_dot"])
@pytest.mark.parametrize("drop", [0.0, 0.25])
def test_pointwise_grads_match_autograd(arch, head, drop):
    """the pointwise sibling (concat-MLP / dot scorer, weighted BCE of seq2vec.py:213-216); ngru / iicgru are
    Seq2VecPaperId's `gru` / `iigru` (paper.py:339-348): 2U-wide user vectors in front of the MLP scorer."""
    if arch in ("ngru", "iicgru") and head == "point_dot":
        pytest.skip("a 2U-wide user vector has no dot scorer")
    rng = np.random.default_rng(5)
    p = O.init_params(rng, arch=arch, dtype=np.float64, head=head, **DIMS)
    for n in ("conv_b", "att_b", "dense_b", "gru_b", "score_b1", "score_b2"):
        if n in p:
            p[n] = rng.normal(0, 0.1, size=p[n].shape)
    B, W, T = 6, 4, 7
    user, hist, cands = make_batch(rng, B, W, T, 1, DIMS["V"], DIMS["N_u"])
    cand = cands[:, 0]
    labels = np.array([1, 0, 0, 1, 0, 0])
    user[2] = user[0]
    gain, K = 1.7, 4
    out = O.point_forward_backward(p, arch, head, user, hist, cand, labels, gain, K, drop_p=drop, seed=9, step=1)
    names = [n for n in p]
    tp = TR.to_t(p, grad_names=names)
    keep1 = keep2 = None
    scale = 1.0
    if drop > 0:
        N_t = B * (W + 1)
        keep1 = O.dropout_keep(O.dropout_key(9, 1, 0), N_t * T * DIMS["D"], drop).reshape(N_t, T, DIMS["D"])
        keep2 = O.dropout_keep(O.dropout_key(9, 1, 1), N_t * T * DIMS["F"], drop).reshape(N_t, T, DIMS["F"])
        scale = 1.0 / (1.0 - drop)
    loss, s, pr = TR.point_loss(tp, arch, head, user, hist, cand, labels, gain, K, keep1, keep2, scale)
    loss.backward()
    assert abs(loss.item() - out["loss"]) < 1e-12
    np.testing.assert_allclose(out["probs"], pr.detach().numpy(), rtol=1e-10, atol=1e-12)
    g = dict(out["grads"])
    g["user_emb"] = O.densify_user_grad(*p["user_emb"].shape, out["user_rows"], out["user_row_grads"], np.float64)
    if arch == "iicgru":
        g["user_emb2"] = O.densify_user_grad(*p["user_emb2"].shape, out["user_rows"], out["user_row_grads2"], np.float64)
    for n in names:
        if n == "word_emb":
            continue                                  # frozen table on this path
        if tp[n].grad is None:
            assert n == "user_emb" and not g[n].any()
            continue
        np.testing.assert_allclose(g[n], tp[n].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=n)
    # the probabilities of predict() are the training-graph probabilities without dropout
    out0 = O.point_forward_backward(p, arch, head, user, hist, cand, labels, gain, K, want_grads=False)
    np.testing.assert_allclose(O.point_predict(p, arch, head, user, hist, cand), out0["probs"], rtol=1e-12)
