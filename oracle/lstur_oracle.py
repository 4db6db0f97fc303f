"""CPU oracle for the LSTUR training path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product path (``mnexp_amd``) never
does; it fails loudly when the HIP extension is missing.

What this is: a NumPy restatement (explicit forward + hand-derived backward +
Keras-style Adam) of the Keras graph the reference builds in

  * news encoder ``cnnatt``      /root/reference/task/paper.py:132-160
  * additive attention pool      /root/reference/models.py:444-492
  * history mask                 /root/reference/models.py:20-30, task/paper.py:644-645
  * user encoder (igru/gru/hgru) /root/reference/task/paper.py:584-633
  * dot scorer + softmax         /root/reference/task/paper.py:443-464
  * loss / optimizer / test head /root/reference/task/paper.py:635-665

PARITY UNPINNED at the Keras boundary: the arithmetic lives in third-party
``keras`` (unpinned in requirements.txt:1; Keras 2.2.4 semantics are assumed)
on TensorFlow 1.x, neither of which is present in /root/reference or
installable here, and the reference ships no golden outputs for this path.
The restatement is therefore cross-checked two independent ways in tests/:
(1) against a torch-CPU autograd re-derivation of the same forward (fp64),
(2) by finite differences.  The integer data path is pinned separately by the
fixtures under tests/golden/.

Layout conventions (shared with the HIP library, see include/lstur.h):
  titles  int32 [N_t, T]   N_t = B*W history titles (b-major) then B*(K+1) candidates
  conv_w  [k, D, F] (Keras Conv1D kernel layout), conv_b [F]
  att_w   [F], att_b [1]
  dense_w [F, U], dense_b [U]
  user_emb [N_u, Ue]
  gru_w [U, 3H], gru_u [H, 3H], gru_b [3H]   gate column order z | r | h
  con_w [H+Ue, U], con_b [U]                  (arch 'gru' only)
"""
from __future__ import annotations

import numpy as np

EPS = 1e-7  # keras.backend.epsilon()

ARCHS = ("igru", "gru", "hgru", "nigru", "ngru", "dgru", "pgru", "vo", "niavg", "niatt", "iigru", "iicgru")
SCORE_MODELS = ("dot", "ddot", "dnn")      # Seq2VecPaperSoftmax._score_model, paper.py:443-458


# --------------------------------------------------------------------------
# counter-based dropout mask (restated identically in csrc/lstur_device.h)
# --------------------------------------------------------------------------
def _fmix32(h):
    h = h.astype(np.uint32, copy=True)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def dropout_key(seed: int, step: int, stream: int) -> int:
    """32-bit key for one (seed, step, stream) triple."""
    with np.errstate(over="ignore"):
        k = np.array([seed & 0xFFFFFFFF], dtype=np.uint32)
        k = _fmix32(k + np.uint32(0x9E3779B9) * np.uint32(stream + 1))
        k = _fmix32(k ^ (np.uint32(step & 0xFFFFFFFF) * np.uint32(0x7FEB352D) + np.uint32(0x165667B1)))
    return int(k[0])


def dropout_keep(key: int, n: int, p: float, start: int = 0):
    """keep[i] for element indices start..start+n-1: one hash per PAIR of elements, its low / high 16 bits against
    floor(p * 2^16)  (csrc/lstur_device.h: keep_elem / keep_quad)."""
    if p <= 0.0:
        return np.ones(n, dtype=bool)
    with np.errstate(over="ignore"):
        idx = np.arange(start, start + n, dtype=np.uint64).astype(np.uint32)
        h = _fmix32((idx >> np.uint32(1)) * np.uint32(0x9E3779B1) + np.uint32(key))
    field = np.where(idx & np.uint32(1), h >> np.uint32(16), h & np.uint32(0xFFFF))
    return field >= np.uint32(int(p * 65536.0))


# --------------------------------------------------------------------------
# initialisers (Keras 2.2.4 defaults for the layers the graph instantiates)
# --------------------------------------------------------------------------
def glorot_uniform(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def orthogonal(rng, rows, cols):
    a = rng.normal(0.0, 1.0, size=(rows, cols))
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == (rows, cols) else vt
    return q[:rows, :cols]


HEADS = ("softmax", "point_mlp", "point_dot")


def init_params(rng, V, D, F, k, U, N_u, arch="igru", word_emb=None, dtype=np.float32, head="softmax",
                encoder="cnnatt", heads=1, score_model="dot"):
    """Random-init parameters with the reference's initialisers
    (conv/dense/att: glorot_uniform, biases zero -- paper.py:146,159, models.py:448-449;
    user table uniform +-0.05 -- paper.py:589; GRU kernel glorot, recurrent orthogonal)."""
    assert arch in ARCHS
    H = U // 2 if arch == "hgru" else U
    Ue = U // 2 if arch == "hgru" else U
    p = {}
    if word_emb is None:
        word_emb = rng.normal(0.0, 0.1, size=(V, D))
        word_emb[0] = 0.0
    p["word_emb"] = word_emb
    if encoder == "mhsa":        # build-defined encoder (no reference counterpart)
        assert F % heads == 0
        p["qkv_w"] = np.concatenate([glorot_uniform(rng, (D, F), D, F) for _ in range(3)], axis=1)   # Wq | Wk | Wv
        p["qkv_b"] = np.zeros(3 * F)
    else:
        p["conv_w"] = glorot_uniform(rng, (k, D, F), k * D, k * F)
        p["conv_b"] = np.zeros(F)
    p["att_w"] = glorot_uniform(rng, (F,), F, 1)
    p["att_b"] = np.zeros(1)
    p["dense_w"] = glorot_uniform(rng, (F, U), F, U)
    p["dense_b"] = np.zeros(U)
    p["user_emb"] = rng.uniform(-0.05, 0.05, size=(N_u, Ue))
    p["gru_w"] = glorot_uniform(rng, (U, 3 * H), U, 3 * H)
    p["gru_u"] = orthogonal(rng, H, 3 * H)
    p["gru_b"] = np.zeros(3 * H)
    if arch in ("gru", "iigru"):
        p["con_w"] = glorot_uniform(rng, (H + Ue, U), H + Ue, U)
        p["con_b"] = np.zeros(U)
    if arch in ("iigru", "iicgru"):   # a second user table behind an igru (paper.py:614-621; pointwise: paper.py:343-348)
        p["user_emb2"] = rng.uniform(-0.05, 0.05, size=(N_u, Ue))
    if arch == "niatt":          # SimpleAttentionMaskSupport over the history (id-less `att`, paper.py:207-209)
        p["uatt_w"] = glorot_uniform(rng, (U,), U, 1)
        p["uatt_b"] = np.zeros(1)
    assert head in HEADS and score_model in SCORE_MODELS
    Uu = 2 * U if arch in ("ngru", "dgru", "iicgru") else U      # width of the user vector (paper.py:600-602, 608-611)
    assert head != "softmax" or score_model != "dot" or Uu == U, "ngru / dgru user vectors are 2U wide: not dot-compatible"
    if score_model == "ddot":    # tanh-Dense on both sides, then dot (paper.py:452-455)
        p["sdu_w"] = glorot_uniform(rng, (Uu, U), Uu, U)
        p["sdu_b"] = np.zeros(U)
        p["sdd_w"] = glorot_uniform(rng, (U, U), U, U)
        p["sdd_b"] = np.zeros(U)
    elif score_model == "dnn":   # concat -> Dense(U, relu) -> Dense(1) (paper.py:448-451)
        p["sn1_w"] = glorot_uniform(rng, (Uu + U, U), Uu + U, U)
        p["sn1_b"] = np.zeros(U)
        p["sn2_w"] = glorot_uniform(rng, (U,), U, 1)
        p["sn2_b"] = np.zeros(1)
    if head == "point_mlp":      # concat_dense / socre_dense of Seq2VecPaper.score_encoder (paper.py:223-227)
        p["score_w1"] = glorot_uniform(rng, (Uu + U, U), Uu + U, U)   # Uu = 2U for Seq2VecPaperId's `gru` (paper.py:339-342)
        p["score_b1"] = np.zeros(U)
        p["score_w2"] = glorot_uniform(rng, (U,), U, 1)
        p["score_b2"] = np.zeros(1)
    p = {n: np.ascontiguousarray(a, dtype=dtype) for n, a in p.items()}
    if encoder == "mhsa":
        p["mhsa_heads"] = np.int64(heads)       # structural, not a weight
    return p


def dense_param_names(arch, word_trainable=False, head="softmax", encoder="cnnatt", score_model="dot"):
    """Order of the flat dense-parameter bucket (same order in the HIP library)."""
    names = ["qkv_w", "qkv_b"] if encoder == "mhsa" else ["conv_w", "conv_b"]
    names += ["att_w", "att_b", "dense_w", "dense_b", "gru_w", "gru_u", "gru_b"]
    if arch in ("gru", "iigru"):
        names += ["con_w", "con_b"]
    if arch == "niatt":
        names += ["uatt_w", "uatt_b"]
    if head == "point_mlp":
        names += ["score_w1", "score_b1", "score_w2", "score_b2"]
    if score_model == "ddot":
        names += ["sdu_w", "sdu_b", "sdd_w", "sdd_b"]
    elif score_model == "dnn":
        names += ["sn1_w", "sn1_b", "sn2_w", "sn2_b"]
    return names


def hard_sigmoid(x):
    return np.clip(0.2 * x + 0.5, 0.0, 1.0)


# --------------------------------------------------------------------------
# news encoder
# --------------------------------------------------------------------------
def _im2col(Epad, T, k):
    n, Tp, D = Epad.shape
    s0, s1, s2 = Epad.strides
    win = np.lib.stride_tricks.as_strided(Epad, shape=(n, T, k, D), strides=(s0, s1, s1, s2), writeable=False)
    return win.reshape(n * T, k * D)


# --------------------------------------------------------------------------
# build-defined news encoder `mhsa` (BASELINE.json configs[4]; SURVEY.md section 8 row a17: the reference has NO
# multi-head self-attention news encoder, so this spec is ours and its parity is against this restatement only):
#   E = Dropout(Embedding(ids));  [Q|K|V] = E . qkv_w + qkv_b   (heads x dh = F columns each)
#   per head: S = Q K^T / sqrt(dh), keys of pad tokens masked, P = softmax_keys(S) (all keys masked -> P = 0)
#   O = P V, heads concatenated, rows of pad tokens zeroed (like paper.py:150-155)
#   then the cnnatt tail unchanged: Masking -> Dropout -> additive attention pool -> Dense(U)
# The HIP path feeds the two attention products (and their gradients) to bf16 MFMA: `bf16_round` marks every
# operand it rounds (accumulation fp32 there, fp64/fp32 here); MHSA_EMULATE_BF16 = False gives the exact math.
# --------------------------------------------------------------------------
MHSA_EMULATE_BF16 = True
MHSA_PROJ_BF16 = True     # the projection products E.W (and E^T.dQKV) also take bf16 operands


def bf16_round(x):
    """Round to nearest-even bfloat16 (restated in csrc/mhsa.hip: f2bf); returns x's dtype."""
    x = np.asarray(x)
    if not MHSA_EMULATE_BF16:
        return x
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32)
    with np.errstate(over="ignore"):
        r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32).astype(x.dtype)


def _proj_round(x):
    return bf16_round(x) if MHSA_PROJ_BF16 else x


def mhsa_fwd(p, E, ids):
    """E [N,T,D] (after dropout), ids [N,T] -> c [N,T,F] and the cache of the backward pass."""
    N, T, D = E.shape
    F = p["qkv_w"].shape[1] // 3
    h = int(p["mhsa_heads"])
    dh = F // h
    dt = E.dtype
    Er = _proj_round(E.reshape(N * T, D))
    qkv = (Er @ _proj_round(p["qkv_w"]) + p["qkv_b"]).reshape(N, T, 3, h, dh)
    Q, K, V = (bf16_round(qkv[:, :, i]) for i in range(3))
    scale = dt.type(1.0 / np.sqrt(dh))
    km = (ids != 0)
    S = np.einsum("nqhd,nkhd->nhqk", Q, K) * scale
    S = np.where(km[:, None, None, :], S, -np.inf)
    m = S.max(axis=-1, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    Pe = np.exp(S - m) * km[:, None, None, :]
    l = Pe.sum(axis=-1, keepdims=True)
    P = np.where(l > 0, Pe / np.where(l > 0, l, 1.0), 0.0).astype(dt)
    Pb = bf16_round(P)
    O = np.einsum("nhqk,nkhd->nqhd", Pb, V).reshape(N, T, F)
    c = O * km[..., None]
    return c, dict(Er=Er, Q=Q, K=K, V=V, P=P, Pb=Pb, scale=scale, km=km)


def mhsa_bwd(p, mc, dc, want_dE=False):
    """dc [N,T,F] (zero on pad rows) -> (qkv_w grad, qkv_b grad, dE or None)."""
    Q, K, V, P, Pb = mc["Q"], mc["K"], mc["V"], mc["P"], mc["Pb"]
    N, T, h, dh = Q.shape
    dO = bf16_round(dc * mc["km"][..., None]).reshape(N, T, h, dh)
    dV = np.einsum("nhqk,nqhd->nkhd", Pb, dO)
    dP = np.einsum("nqhd,nkhd->nhqk", dO, V)
    delta = (P * dP).sum(axis=-1, keepdims=True)
    dS = bf16_round(P * (dP - delta) * mc["scale"])
    dQ = np.einsum("nhqk,nkhd->nqhd", dS, K)
    dK = np.einsum("nhqk,nqhd->nkhd", dS, Q)
    dqkv = np.stack([dQ, dK, dV], axis=2).reshape(N * T, 3 * h * dh)
    dqr = _proj_round(dqkv)             # the HIP path stores dQ|dK|dV rows as bf16; the bias gradient sums those
    gb = dqr.sum(axis=0)
    gw = mc["Er"].T @ dqr
    dE = (dqr @ _proj_round(p["qkv_w"]).T).reshape(N, T, -1) if want_dE else None
    return gw, gb, dE


def news_encoder_fwd(p, titles, drop_p=0.0, keys=(0, 0), training=True, chunk=4096):
    """titles int [N_t,T] -> doc vectors [N_t,U]; returns (v, cache)."""
    mhsa = "qkv_w" in p
    dt = p["att_w"].dtype
    ids = np.asarray(titles).astype(np.int64)
    N, T = ids.shape
    F = p["att_w"].shape[0]
    use_drop = training and drop_p > 0.0
    scale = dt.type(1.0 / (1.0 - drop_p)) if use_drop else dt.type(1.0)
    mcache = None
    if mhsa:
        E = p["word_emb"][ids].astype(dt)
        if use_drop:
            E = E * dropout_keep(keys[0], N * T * E.shape[-1], drop_p).reshape(E.shape) * scale
        c, mcache = mhsa_fwd(p, E, ids)
    else:
        k, D, _ = p["conv_w"].shape
        pl = (k - 1) // 2
        Wm = p["conv_w"].reshape(k * D, F)
        c = np.empty((N, T, F), dtype=dt)
        for s in range(0, N, chunk):
            e = min(N, s + chunk)
            E = p["word_emb"][ids[s:e]].astype(dt)  # gather, pad id 0 is gathered too
            if use_drop:
                keep = dropout_keep(keys[0], (e - s) * T * D, drop_p, start=s * T * D).reshape(e - s, T, D)
                E = E * keep * scale
            Epad = np.zeros((e - s, T + k - 1, D), dtype=dt)
            Epad[:, pl:pl + T] = E
            cc = _im2col(Epad, T, k) @ Wm + p["conv_b"]
            c[s:e] = np.maximum(cc, 0).reshape(e - s, T, F)
        c *= (ids != 0)[..., None]                      # paper.py:150-155
    tmask = np.any(c != 0, axis=-1)                 # keras Masking(0.) paper.py:157
    if use_drop:
        keep2 = dropout_keep(keys[1], N * T * F, drop_p).reshape(N, T, F)
        x = c * keep2 * scale
    else:
        keep2 = None
        x = c
    a = np.tanh(x @ p["att_w"] + p["att_b"][0])     # models.py:474-477
    e_ = np.exp(a) * tmask                          # models.py:480
    S = e_.sum(axis=1, keepdims=True) + dt.type(EPS)
    w = e_ / S                                      # models.py:485
    pooled = np.einsum("ntf,nt->nf", x, w)
    v = pooled @ p["dense_w"] + p["dense_b"]        # paper.py:159
    cache = dict(ids=ids, c=c, tmask=tmask, keep2=keep2, x=x, a=a, w=w, pooled=pooled,
                 scale=scale, drop_p=drop_p if use_drop else 0.0, keys=keys, mhsa=mcache)
    return v, cache


def news_encoder_bwd(p, cache, dv, word_trainable=False, chunk=4096):
    """dv [N_t,U] -> grads dict for conv (or qkv) /att/dense (+ word_emb when trainable)."""
    dt = p["att_w"].dtype
    ids, c, x, a, w = cache["ids"], cache["c"], cache["x"], cache["a"], cache["w"]
    N, T = ids.shape
    g = {}
    g["dense_w"] = cache["pooled"].T @ dv
    g["dense_b"] = dv.sum(axis=0)
    dpooled = dv @ p["dense_w"].T                   # [N,F]
    dw = np.einsum("nf,ntf->nt", dpooled, x)
    # w = e/(S+eps): de_t = (dw_t - sum_t' dw_t' w_t')/(S+eps); da_t = de_t*e_t = (..)*w_t
    da = (dw - (dw * w).sum(axis=1, keepdims=True)) * w
    ds = da * (1.0 - a * a)
    dx = dpooled[:, None, :] * w[..., None] + ds[..., None] * p["att_w"][None, None, :]
    g["att_w"] = np.einsum("nt,ntf->f", ds, x)
    g["att_b"] = np.array([ds.sum()], dtype=dt)
    if cache["keep2"] is not None:
        dx = dx * cache["keep2"] * cache["scale"]
    use_drop = cache["drop_p"] > 0.0
    if cache["mhsa"] is not None:                   # no ReLU behind the self-attention: only the pad-row mask
        dc = dx * (ids != 0)[..., None]
        g["qkv_w"], g["qkv_b"], dE = mhsa_bwd(p, cache["mhsa"], dc, want_dE=word_trainable)
        if word_trainable:
            if use_drop:
                dE = dE * dropout_keep(cache["keys"][0], dE.size, cache["drop_p"]).reshape(dE.shape) * cache["scale"]
            g_word = np.zeros_like(p["word_emb"])
            np.add.at(g_word, ids.reshape(-1), dE.reshape(N * T, -1))
            g["word_emb"] = g_word
        return g
    k, D, F = p["conv_w"].shape
    pl = (k - 1) // 2
    dc = dx * (c > 0)                               # ReLU' ; pad-masked / masked rows have c == 0
    g["conv_b"] = dc.sum(axis=(0, 1))
    gW = np.zeros((k * D, F), dtype=dt)
    if word_trainable:
        g_word = np.zeros_like(p["word_emb"])
        Wm = p["conv_w"].reshape(k * D, F)
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        E = p["word_emb"][ids[s:e]].astype(dt)
        if use_drop:
            keep = dropout_keep(cache["keys"][0], (e - s) * T * D, cache["drop_p"], start=s * T * D).reshape(e - s, T, D)
            E = E * keep * cache["scale"]
        Epad = np.zeros((e - s, T + k - 1, D), dtype=dt)
        Epad[:, pl:pl + T] = E
        dcf = dc[s:e].reshape((e - s) * T, F)
        gW += _im2col(Epad, T, k).T @ dcf
        if word_trainable:
            dcol = (dcf @ Wm.T).reshape(e - s, T, k, D)
            dEpad = np.zeros((e - s, T + k - 1, D), dtype=dt)
            for j in range(k):
                dEpad[:, j:j + T] += dcol[:, :, j]
            dE = dEpad[:, pl:pl + T]
            if use_drop:
                dE = dE * keep * cache["scale"]
            np.add.at(g_word, ids[s:e].reshape(-1), dE.reshape(-1, D))
    g["conv_w"] = gW.reshape(k, D, F)
    if word_trainable:
        g["word_emb"] = g_word
    return g


# --------------------------------------------------------------------------
# masked GRU (Keras 2.2.4: gates z|r|h, hard_sigmoid / tanh, reset_after=False)
# --------------------------------------------------------------------------
def gru_fwd(p, x, gmask, h0):
    """x [B,W,U], gmask bool [B,W], h0 [B,H] -> last output [B,H]."""
    dt = x.dtype
    B, W, _ = x.shape
    H = p["gru_u"].shape[0]
    xw = x.reshape(B * W, -1) @ p["gru_w"] + p["gru_b"]
    xw = xw.reshape(B, W, 3 * H)
    Uz, Ur, Uh = p["gru_u"][:, :H], p["gru_u"][:, H:2 * H], p["gru_u"][:, 2 * H:]
    hs = np.empty((B, W + 1, H), dtype=dt)
    hs[:, 0] = h0
    zs = np.empty((B, W, H), dtype=dt)
    rs = np.empty((B, W, H), dtype=dt)
    hhs = np.empty((B, W, H), dtype=dt)
    h = h0
    for t in range(W):
        z = hard_sigmoid(xw[:, t, :H] + h @ Uz)
        r = hard_sigmoid(xw[:, t, H:2 * H] + h @ Ur)
        hh = np.tanh(xw[:, t, 2 * H:] + (r * h) @ Uh)
        hn = z * h + (1 - z) * hh
        m = gmask[:, t][:, None]
        h = np.where(m, hn, h)                      # K.rnn: masked step carries state and output
        zs[:, t], rs[:, t], hhs[:, t], hs[:, t + 1] = z, r, hh, h
    return h, dict(x=x, gmask=gmask, hs=hs, zs=zs, rs=rs, hhs=hhs, xw=xw)


def _hsig_grad(pre):
    # tf.clip_by_value passes the gradient where 0 <= 0.2x+0.5 <= 1 (inclusive)
    y = 0.2 * pre + 0.5
    return np.where((y >= 0.0) & (y <= 1.0), 0.2, 0.0)


def gru_bwd(p, cache, dh_last):
    dt = dh_last.dtype
    x, gmask, hs, zs, rs, hhs, xw = (cache[n] for n in ("x", "gmask", "hs", "zs", "rs", "hhs", "xw"))
    B, W, _ = x.shape
    H = p["gru_u"].shape[0]
    Uz, Ur, Uh = p["gru_u"][:, :H], p["gru_u"][:, H:2 * H], p["gru_u"][:, 2 * H:]
    dgates = np.zeros((B, W, 3 * H), dtype=dt)
    dU = np.zeros_like(p["gru_u"])
    dh = dh_last.copy()
    for t in range(W - 1, -1, -1):
        m = gmask[:, t][:, None]
        hp = hs[:, t]
        z, r, hh = zs[:, t], rs[:, t], hhs[:, t]
        dhn = np.where(m, dh, 0.0)
        dh_carry = np.where(m, 0.0, dh)
        dz = dhn * (hp - hh)
        dhh = dhn * (1 - z)
        dpre_h = dhh * (1 - hh * hh)
        drh = dpre_h @ Uh.T
        dr = drh * hp
        pre_z = xw[:, t, :H] + hp @ Uz
        pre_r = xw[:, t, H:2 * H] + hp @ Ur
        dpre_z = dz * _hsig_grad(pre_z)
        dpre_r = dr * _hsig_grad(pre_r)
        dgates[:, t, :H] = dpre_z
        dgates[:, t, H:2 * H] = dpre_r
        dgates[:, t, 2 * H:] = dpre_h
        dU[:, :H] += hp.T @ dpre_z
        dU[:, H:2 * H] += hp.T @ dpre_r
        dU[:, 2 * H:] += (r * hp).T @ dpre_h
        dh = dh_carry + dhn * z + drh * r + dpre_z @ Uz.T + dpre_r @ Ur.T
    dg = dgates.reshape(B * W, 3 * H)
    g = dict(gru_w=x.reshape(B * W, -1).T @ dg, gru_b=dg.sum(axis=0), gru_u=dU)
    dx = (dg @ p["gru_w"].T).reshape(x.shape)
    return g, dx, dh


# --------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------
def split_titles(v, B, W, K1):
    hv = v[:B * W].reshape(B, W, -1)
    dv = v[B * W:].reshape(B, K1, -1)
    return hv, dv


def user_encoder_fwd(p, arch, user, hv, hist_ids, uvec_keep=None, keep_scale=2.0):
    """paper.py:584-633 for arch in igru|gru|hgru|nigru|ngru|dgru; hv [B,W,U] raw doc vectors of the history.
    uvec_keep (training): bool [B], whole-vector dropout of the user embedding, kept rows scaled by `keep_scale` --
    dgru's Dropout(0.5, noise_shape=(None, 1)) of paper.py:610, or the `id_keep` masking of cook.py:139-142
    (idx_vec * Dropout(1 - id_keep)(idx_mask): scale 1 / id_keep); iigru: the same rows of both tables."""
    hmask = np.any(hist_ids != 0, axis=-1)           # models.ComputeMasking, paper.py:644
    hvm = hv * hmask[..., None]                      # paper.py:645
    gmask = np.any(hvm != 0, axis=-1)                # Masking() paper.py:592
    uvec = p["user_emb"][user]
    uvec2 = p["user_emb2"][user] if arch in ("iigru", "iicgru") else None
    if uvec_keep is not None:
        uvec = uvec * uvec_keep[:, None] * hv.dtype.type(keep_scale)
        if uvec2 is not None:
            uvec2 = uvec2 * uvec_keep[:, None] * hv.dtype.type(keep_scale)
    H = p["gru_u"].shape[0]
    if arch == "igru":
        h, gc = gru_fwd(p, hvm, gmask, uvec)
        u = h
        extra = None
    elif arch == "nigru":                           # paper.py:627-628 / 206-208: no user embedding in the graph
        h, gc = gru_fwd(p, hvm, gmask, np.zeros((len(user), H), dtype=hv.dtype))
        u = h
        extra = None
    elif arch == "pgru":                            # paper.py:624-626: GRU(U)(x) + user_vec
        h, gc = gru_fwd(p, hvm, gmask, np.zeros((len(user), H), dtype=hv.dtype))
        u = h + uvec
        extra = None
    elif arch == "iigru":                           # paper.py:614-621: Dense(U)(concat[igru(x, user_vec), user_vec2])
        h, gc = gru_fwd(p, hvm, gmask, uvec)
        extra = np.concatenate([h, uvec2], axis=1)
        u = extra @ p["con_w"] + p["con_b"]
    elif arch == "iicgru":                          # Seq2VecPaperId's `iigru` (paper.py:343-348): the same concat, NO Dense
        h, gc = gru_fwd(p, hvm, gmask, uvec)
        u = extra = np.concatenate([h, uvec2], axis=1)
    elif arch == "vo":                              # paper.py:622-623: the user embedding alone, the history is unused
        u, gc, extra = uvec, None, None
    elif arch == "niavg":                           # paper.py:629-630 + models.py:433-435: masked mean, no user id
        cnt = gmask.sum(axis=1, keepdims=True).astype(hv.dtype)
        u = hvm.sum(axis=1) / (cnt + hv.dtype.type(EPS))
        gc, extra = (gmask, cnt), None
    elif arch == "niatt":                           # paper.py:207-209 + models.py:474-489: additive attention, no user id
        a = np.tanh(hvm @ p["uatt_w"] + p["uatt_b"][0])
        e_ = np.exp(a) * gmask
        wgt = e_ / (e_.sum(axis=1, keepdims=True) + hv.dtype.type(EPS))
        u = np.einsum("bwu,bw->bu", hvm, wgt)
        gc, extra = (hvm, a, wgt), None
    else:
        h, gc = gru_fwd(p, hvm, gmask, np.zeros((len(user), H), dtype=hv.dtype))
        cat = np.concatenate([h, uvec], axis=1)
        if arch == "gru":
            u = cat @ p["con_w"] + p["con_b"]
        else:  # hgru (H = Ue = U/2), ngru / dgru (H = Ue = U: a 2U-wide user vector)
            u = cat
        extra = cat
    return u, dict(gc=gc, hmask=hmask, user=user, cat=extra, arch=arch, uvec_keep=uvec_keep, keep_scale=keep_scale)


def user_encoder_bwd(p, cache, du):
    arch, gc = cache["arch"], cache["gc"]
    H = p["gru_u"].shape[0]
    g = {}
    if arch == "igru":
        gg, dx, dh0 = gru_bwd(p, gc, du)
        duser = dh0
    elif arch == "nigru":
        gg, dx, _ = gru_bwd(p, gc, du)
        duser = np.zeros_like(du)
    elif arch == "pgru":
        gg, dx, _ = gru_bwd(p, gc, du)
        duser = du
    elif arch == "iigru":
        g["con_w"] = cache["cat"].T @ du
        g["con_b"] = du.sum(axis=0)
        dcat = du @ p["con_w"].T
        gg, dx, dh0 = gru_bwd(p, gc, np.ascontiguousarray(dcat[:, :H]))
        duser = dh0
        cache["duser2"] = dcat[:, H:]               # row gradients of the second user table
    elif arch == "iicgru":
        gg, dx, dh0 = gru_bwd(p, gc, np.ascontiguousarray(du[:, :H]))
        duser = dh0
        cache["duser2"] = du[:, H:]
    elif arch in ("vo", "niavg", "niatt"):          # no GRU in the graph: its parameters get no gradient
        gg = {n: np.zeros_like(p[n]) for n in ("gru_w", "gru_u", "gru_b")}
        if arch == "niatt":
            x, a, wgt = gc
            dw = np.einsum("bu,bwu->bw", du, x)
            da = (dw - (dw * wgt).sum(axis=1, keepdims=True)) * wgt
            ds = da * (1.0 - a * a)
            dx = du[:, None, :] * wgt[..., None] + ds[..., None] * p["uatt_w"][None, None, :]
            gg["uatt_w"] = np.einsum("bw,bwu->u", ds, x)
            gg["uatt_b"] = np.array([ds.sum()], dtype=du.dtype)
            duser = np.zeros_like(du)
        elif arch == "vo":
            dx = np.zeros(cache["hmask"].shape + (du.shape[1],), dtype=du.dtype)
            duser = du
        else:
            gmask, cnt = gc
            dx = gmask[..., None] * (du / (cnt + du.dtype.type(EPS)))[:, None, :]
            duser = np.zeros_like(du)
    else:
        if arch == "gru":
            g["con_w"] = cache["cat"].T @ du
            g["con_b"] = du.sum(axis=0)
            dcat = du @ p["con_w"].T
        else:
            dcat = du
        gg, dx, _ = gru_bwd(p, gc, np.ascontiguousarray(dcat[:, :H]))
        duser = dcat[:, H:]
    if cache["uvec_keep"] is not None:              # the gradient passes the same row mask and scale
        duser = duser * cache["uvec_keep"][:, None] * du.dtype.type(cache["keep_scale"])
        if "duser2" in cache:
            cache["duser2"] = cache["duser2"] * cache["uvec_keep"][:, None] * du.dtype.type(cache["keep_scale"])
    g.update(gg)
    dhv = dx * cache["hmask"][..., None]
    return g, dhv, duser


def softmax_ce(z):
    """softmax -> Keras categorical_crossentropy on probabilities with label [1,0,..,0]
    (paper.py:463, 655-659); returns per-sample loss, probabilities and dL/dz (per sample)."""
    dt = z.dtype
    zs = z - z.max(axis=1, keepdims=True)
    ez = np.exp(zs)
    pr = ez / ez.sum(axis=1, keepdims=True)
    S = pr.sum(axis=1, keepdims=True)
    q = pr / S
    cq = np.clip(q, dt.type(EPS), dt.type(1.0 - EPS))
    loss = -np.log(cq[:, 0])
    inside = (q[:, 0] >= dt.type(EPS)) & (q[:, 0] <= dt.type(1.0 - EPS))
    gq = np.zeros_like(pr)
    gq[:, 0] = np.where(inside, -1.0 / cq[:, 0], 0.0)
    dp = gq / S - (gq * pr).sum(axis=1, keepdims=True) / (S * S)
    dz = pr * (dp - (dp * pr).sum(axis=1, keepdims=True))
    return loss, pr, dz


def score_model_of(p):
    return "ddot" if "sdu_w" in p else ("dnn" if "sn1_w" in p else "dot")


def listwise_score_fwd(p, u, dv):
    """score_model applied to every candidate (paper.py:443-464): u [B,Uu], dv [B,K1,U] -> z [B,K1]."""
    sm = score_model_of(p)
    if sm == "dot":
        return np.einsum("bu,bju->bj", u, dv), None
    if sm == "ddot":
        uh = np.tanh(u @ p["sdu_w"] + p["sdu_b"])
        dh = np.tanh(dv @ p["sdd_w"] + p["sdd_b"])
        return np.einsum("bu,bju->bj", uh, dh), (uh, dh)
    B, K1, U = dv.shape
    cat = np.concatenate([np.repeat(u[:, None, :], K1, axis=1), dv], axis=2).reshape(B * K1, -1)
    pre = cat @ p["sn1_w"] + p["sn1_b"]
    hid = np.maximum(pre, 0.0)
    z = (hid @ p["sn2_w"] + p["sn2_b"][0]).reshape(B, K1)
    return z, (cat, pre, hid)


def listwise_score_bwd(p, u, dv, cache, dz):
    """dz [B,K1] -> (du [B,Uu], ddv [B,K1,U], grads of the score model's own parameters)."""
    sm = score_model_of(p)
    g = {}
    if sm == "dot":
        return np.einsum("bj,bju->bu", dz, dv), dz[..., None] * u[:, None, :], g
    B, K1, U = dv.shape
    if sm == "ddot":
        uh, dh = cache
        duh = np.einsum("bj,bju->bu", dz, dh) * (1.0 - uh * uh)
        ddh = (dz[..., None] * uh[:, None, :]) * (1.0 - dh * dh)
        g["sdu_w"] = u.T @ duh
        g["sdu_b"] = duh.sum(axis=0)
        g["sdd_w"] = dv.reshape(B * K1, U).T @ ddh.reshape(B * K1, U)
        g["sdd_b"] = ddh.sum(axis=(0, 1))
        return duh @ p["sdu_w"].T, ddh @ p["sdd_w"].T, g
    cat, pre, hid = cache
    ds = dz.reshape(B * K1)
    g["sn2_w"] = hid.T @ ds
    g["sn2_b"] = np.array([ds.sum()], dtype=u.dtype)
    dhid = ds[:, None] * p["sn2_w"][None, :] * (pre > 0)
    g["sn1_w"] = cat.T @ dhid
    g["sn1_b"] = dhid.sum(axis=0)
    dcat = (dhid @ p["sn1_w"].T).reshape(B, K1, -1)
    Uu = u.shape[1]
    return dcat[:, :, :Uu].sum(axis=1), dcat[:, :, Uu:], g


def forward_backward(p, arch, user, hist, cands, drop_p=0.0, seed=0, step=0, word_trainable=False,
                     want_grads=True, stream_base=0, id_keep=1.0):
    """One training step's loss + gradients.
    user int [B]; hist int [B,W,T]; cands int [B,K+1,T] (candidate 0 is the positive).
    `stream_base`: first of the three dropout streams (embeddings, conv outputs, user-embedding rows) this call draws from.
    `id_keep` < 1: a sample keeps its user embedding with this probability (cook.py:139-142); dgru always masks with 0.5.
    Returns dict(loss, acc, probs, logits, grads{name: array}, user_rows, user_row_grads)."""
    user = np.asarray(user).astype(np.int64).reshape(-1)
    hist = np.asarray(hist).astype(np.int64)
    cands = np.asarray(cands).astype(np.int64)
    B, W, T = hist.shape
    K1 = cands.shape[1]
    titles = np.concatenate([hist.reshape(B * W, T), cands.reshape(B * K1, T)], axis=0)
    keys = (dropout_key(seed, step, stream_base), dropout_key(seed, step, stream_base + 1))
    v, nc = news_encoder_fwd(p, titles, drop_p, keys, training=True)
    hv, dv = split_titles(v, B, W, K1)
    row_p = 0.5 if arch == "dgru" else (1.0 - id_keep if 0.0 < id_keep < 1.0 and arch not in ("nigru", "niavg", "niatt") else 0.0)
    keep_u = dropout_keep(dropout_key(seed, step, stream_base + 2), B, row_p) if row_p > 0.0 else None   # paper.py:610
    u, uc = user_encoder_fwd(p, arch, user, hv, hist, keep_u, 1.0 / (1.0 - row_p) if row_p > 0.0 else 1.0)
    z, sc = listwise_score_fwd(p, u, dv)
    loss_i, pr, dz = softmax_ce(z)
    out = dict(loss=float(loss_i.mean()), acc=float((pr.argmax(axis=1) == 0).mean()),
               probs=pr, logits=z, user_vec=u, doc_vec=v)
    if not want_grads:
        return out
    dz = dz / v.dtype.type(B)
    du, ddv, gs = listwise_score_bwd(p, u, dv, sc, dz)
    g, dhv, duser = user_encoder_bwd(p, uc, du)
    g.update(gs)
    dvv = np.concatenate([dhv.reshape(B * W, -1), ddv.reshape(B * K1, -1)], axis=0)
    g.update(news_encoder_bwd(p, nc, dvv, word_trainable))
    out["grads"] = g
    out["user_rows"] = user
    out["user_row_grads"] = duser
    if arch in ("iigru", "iicgru"):
        out["user_row_grads2"] = uc["duser2"]
    return out


LANE_MIN_BATCH = 32      # csrc/lstur_capi.hip: kPipeMin


def lane_split(n):
    """Samples of lane 0 of the library's two-lane training step (csrc/lstur_capi.hip: lane0_share); batches below
    LANE_MIN_BATCH are not cut."""
    return max(16, (n // 2) // 16 * 16) if n >= LANE_MIN_BATCH else n


def forward_backward_lanes(p, arch, user, hist, cands, drop_p=0.0, seed=0, step=0, id_keep=1.0):
    """The same step the way the library runs a batch of >= LANE_MIN_BATCH samples (csrc/lstur_capi.hip:
    train_two_lanes): two sample ranges, each encoded in its OWN [history | candidates] title layout -- lane 1's dropout
    masks come from streams 16 + (0, 1, 2) at the element positions of that layout -- the loss is the mean over the whole
    batch, gradients add.  Mathematically the step of `forward_backward` on the whole batch; only WHICH elements the
    dropout hash drops differs, so parity tests of large batches go through this function."""
    user = np.asarray(user).reshape(-1)
    n = len(user)
    nA = lane_split(n)
    if nA >= n:
        return forward_backward(p, arch, user, hist, cands, drop_p, seed, step, id_keep=id_keep)
    outs, ns = [], []
    for lane, (a, b) in enumerate(((0, nA), (nA, n))):
        outs.append(forward_backward(p, arch, user[a:b], np.asarray(hist)[a:b], np.asarray(cands)[a:b], drop_p, seed, step,
                                     stream_base=16 * lane, id_keep=id_keep))
        ns.append(b - a)
    wgt = [x / float(n) for x in ns]
    out = dict(loss=sum(w * o["loss"] for w, o in zip(wgt, outs)), acc=sum(w * o["acc"] for w, o in zip(wgt, outs)),
               probs=np.concatenate([o["probs"] for o in outs]), logits=np.concatenate([o["logits"] for o in outs]),
               user_rows=np.concatenate([o["user_rows"] for o in outs]),
               # each lane's gradients are those of ITS mean loss: weight by its share of the batch
               user_row_grads=np.concatenate([w * o["user_row_grads"] for w, o in zip(wgt, outs)]),
               grads={k: sum(w * o["grads"][k] for w, o in zip(wgt, outs)) for k in outs[0]["grads"]})
    if "user_row_grads2" in outs[0]:
        out["user_row_grads2"] = np.concatenate([w * o["user_row_grads2"] for w, o in zip(wgt, outs)])
    return out


def predict(p, arch, user, hist, cand):
    """test_model (paper.py:661-665): sigmoid(u . d) for one candidate per row.
    user [n], hist [n,W,T], cand [n,T] -> [n]"""
    user = np.asarray(user).astype(np.int64).reshape(-1)
    hist = np.asarray(hist).astype(np.int64)
    cand = np.asarray(cand).astype(np.int64)
    n, W, T = hist.shape
    titles = np.concatenate([hist.reshape(n * W, T), cand.reshape(n, T)], axis=0)
    v, _ = news_encoder_fwd(p, titles, training=False)
    hv, dv = split_titles(v, n, W, 1)
    u, _ = user_encoder_fwd(p, arch, user, hv, hist)
    z, _ = listwise_score_fwd(p, u, dv)              # sigmoid(score_model([u, d])), paper.py:661-665
    return 1.0 / (1.0 + np.exp(-z[:, 0]))


# --------------------------------------------------------------------------
# pointwise sibling: Seq2VecPaper / Seq2VecPaperDot / Seq2VecPaperId
# --------------------------------------------------------------------------
def point_score_fwd(p, head, u, d):
    """One candidate per row.  point_mlp: Seq2VecPaper.score_encoder (paper.py:223-227): concat -> Dense(U, relu)
    -> Dense(1, sigmoid); point_dot: Seq2VecPaperDot.score_encoder (paper.py:258-262): sigmoid(u . d)."""
    if head == "point_dot":
        s = np.einsum("bu,bu->b", u, d)
        cache = None
    else:
        cat = np.concatenate([u, d], axis=1)
        pre = cat @ p["score_w1"] + p["score_b1"]
        hid = np.maximum(pre, 0.0)
        s = hid @ p["score_w2"] + p["score_b2"][0]
        cache = (cat, pre, hid)
    return s, 1.0 / (1.0 + np.exp(-s)), cache


def point_score_bwd(p, head, u, d, cache, ds):
    g = {}
    if head == "point_dot":
        return g, ds[:, None] * d, ds[:, None] * u
    cat, pre, hid = cache
    g["score_w2"] = hid.T @ ds
    g["score_b2"] = np.array([ds.sum()], dtype=ds.dtype)
    dhid = ds[:, None] * p["score_w2"][None, :] * (pre > 0)
    g["score_w1"] = cat.T @ dhid
    g["score_b1"] = dhid.sum(axis=0)
    dcat = dhid @ p["score_w1"].T
    U = u.shape[1]
    return g, dcat[:, :U], dcat[:, U:]


def weighted_bce(prob, y, gain, K):
    """Seq2Vec.loss (seq2vec.py:213-216): -0.5 (1+K) mean(y log(p+1e-8) gain + (1-y) log(1-p+1e-8) / K).
    Returns per-sample loss terms (their mean is the loss) and dL_i/dp_i (not yet divided by the batch)."""
    dt = prob.dtype.type
    c = dt(0.5 * (1 + K))
    e = dt(1e-8)
    li = -c * (y * np.log(prob + e) * dt(gain) + (1 - y) * np.log(1 - prob + e) / dt(K))
    dp = -c * (y * dt(gain) / (prob + e) - (1 - y) / (dt(K) * (1 - prob + e)))
    return li, dp


def point_forward_backward(p, arch, head, user, hist, cand, labels, gain=1.0, K=4, drop_p=0.0, seed=0, step=0,
                           want_grads=True):
    """One pointwise training step (Seq2VecPaperId._build_model, paper.py:356-383).
    user int [B]; hist int [B,W,T]; cand int [B,T]; labels {0,1} [B]."""
    user = np.asarray(user).astype(np.int64).reshape(-1)
    hist = np.asarray(hist).astype(np.int64)
    cand = np.asarray(cand).astype(np.int64)
    B, W, T = hist.shape
    titles = np.concatenate([hist.reshape(B * W, T), cand.reshape(B, T)], axis=0)
    keys = (dropout_key(seed, step, 0), dropout_key(seed, step, 1))
    v, nc = news_encoder_fwd(p, titles, drop_p, keys, training=True)
    hv, dv = split_titles(v, B, W, 1)
    d = dv[:, 0]
    u, uc = user_encoder_fwd(p, arch, user, hv, hist)
    y = np.asarray(labels).astype(v.dtype).reshape(-1)
    s, prob, cache = point_score_fwd(p, head, u, d)
    li, dp = weighted_bce(prob, y, gain, K)
    out = dict(loss=float(li.mean()), acc=float(((prob > 0.5) == (y > 0.5)).mean()), probs=prob, logits=s,
               user_vec=u, doc_vec=v)
    if not want_grads:
        return out
    ds = dp * prob * (1 - prob) / v.dtype.type(B)
    gs, du, dd = point_score_bwd(p, head, u, d, cache, ds)
    g, dhv, duser = user_encoder_bwd(p, uc, du)
    g.update(gs)
    dvv = np.concatenate([dhv.reshape(B * W, -1), dd], axis=0)
    g.update(news_encoder_bwd(p, nc, dvv, False))
    out["grads"] = g
    out["user_rows"] = user
    out["user_row_grads"] = duser
    if "duser2" in uc:
        out["user_row_grads2"] = uc["duser2"]
    return out


def point_predict(p, arch, head, user, hist, cand):
    """model.predict of the pointwise model (seq2vec.py:206): the scorer's sigmoid output, [n]."""
    user = np.asarray(user).astype(np.int64).reshape(-1)
    hist = np.asarray(hist).astype(np.int64)
    cand = np.asarray(cand).astype(np.int64)
    n, W, T = hist.shape
    titles = np.concatenate([hist.reshape(n * W, T), cand.reshape(n, T)], axis=0)
    v, _ = news_encoder_fwd(p, titles, training=False)
    hv, dv = split_titles(v, n, W, 1)
    u, _ = user_encoder_fwd(p, arch, user, hv, hist)
    return point_score_fwd(p, head, u, dv[:, 0])[1]


def point_train_step(p, arch, head, opt, user, hist, cand, labels, gain=1.0, K=4, drop_p=0.0, seed=0, step=0):
    out = point_forward_backward(p, arch, head, user, hist, cand, labels, gain, K, drop_p, seed, step)
    g = out["grads"]
    g["user_emb"] = densify_user_grad(*p["user_emb"].shape, out["user_rows"], out["user_row_grads"],
                                      p["user_emb"].dtype)
    if "user_row_grads2" in out:
        g["user_emb2"] = densify_user_grad(*p["user_emb2"].shape, out["user_rows"], out["user_row_grads2"],
                                           p["user_emb2"].dtype)
    opt.step(p, g)
    return out


# --------------------------------------------------------------------------
# Keras-2.2.4 Adam (eps outside sqrt(v), bias correction folded into lr_t,
# DENSE update of the embedding table: TF-1 densifies the IndexedSlices grad)
# --------------------------------------------------------------------------
class Adam:
    def __init__(self, params, names, lr=1e-3, beta1=0.9, beta2=0.999, eps=EPS):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.names = list(names)
        self.m = {n: np.zeros_like(params[n]) for n in self.names}
        self.v = {n: np.zeros_like(params[n]) for n in self.names}

    def lr_t(self, dt=np.float32):
        """Evaluated in the parameter dtype like the TF graph (fp32 for Keras)."""
        t = dt(self.t)
        return dt(self.lr) * (np.sqrt(dt(1.0) - np.power(dt(self.b2), t)) / (dt(1.0) - np.power(dt(self.b1), t)))

    def step(self, params, grads):
        self.t += 1
        for n in self.names:
            dt = params[n].dtype.type
            g = grads[n].astype(params[n].dtype)
            lr_t = self.lr_t(dt)
            self.m[n] = dt(self.b1) * self.m[n] + dt(1.0 - self.b1) * g
            self.v[n] = dt(self.b2) * self.v[n] + dt(1.0 - self.b2) * (g * g)
            params[n] = params[n] - lr_t * self.m[n] / (np.sqrt(self.v[n]) + dt(self.eps))


def densify_user_grad(N_u, Ue, rows, row_grads, dtype):
    """Sum duplicate rows in ascending sample order (the order the HIP library uses)."""
    g = np.zeros((N_u, Ue), dtype=dtype)
    for r, gr in zip(rows, row_grads):
        g[r] += gr
    return g


def train_step(p, arch, opt, user, hist, cands, drop_p=0.0, seed=0, step=0, word_trainable=False, id_keep=1.0):
    """forward/backward + Adam over dense params and the (densified) user table (+ the word table when
    --textual-embedding-trainable: `opt` must then list 'word_emb'; its IndexedSlices gradient is densified too)."""
    out = forward_backward(p, arch, user, hist, cands, drop_p, seed, step, word_trainable=word_trainable, id_keep=id_keep)
    g = out["grads"]
    g["user_emb"] = densify_user_grad(*p["user_emb"].shape, out["user_rows"], out["user_row_grads"],
                                      p["user_emb"].dtype)
    if "user_row_grads2" in out:                    # iigru: the second user table
        g["user_emb2"] = densify_user_grad(*p["user_emb2"].shape, out["user_rows"], out["user_row_grads2"],
                                           p["user_emb2"].dtype)
    opt.step(p, g)
    return out
