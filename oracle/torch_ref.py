"""Independent torch-CPU (autograd) re-derivation of the LSTUR forward pass.

Test infrastructure only (never imported by the product path): cross-checks the hand-written backward of
oracle/lstur_oracle.py ("cross-checked two independent ways") and is the torch-CPU fp32 leg of
bench.py's `cpu_baseline` (torch-CPU, all cores).
Written against the Keras-2.2.4 semantics listed there, using torch ops
(F.conv1d, autograd) rather than the oracle's im2col / explicit gradients.
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-7


def to_t(p, dtype=torch.float64, grad_names=()):
    out = {}
    for n, a in p.items():
        t = torch.tensor(np.asarray(a), dtype=dtype)
        if n in grad_names:
            t.requires_grad_(True)
        out[n] = t
    return out


def news_encoder(p, titles, keep1=None, keep2=None, scale=1.0):
    ids = torch.as_tensor(np.asarray(titles), dtype=torch.long)
    N, T = ids.shape
    E = p["word_emb"][ids]                                  # [N,T,D]
    if keep1 is not None:
        E = E * torch.as_tensor(keep1, dtype=E.dtype) * scale
    if "qkv_w" in p:                                        # build-defined mhsa encoder (exact math, no bf16 rounding)
        h = int(p["mhsa_heads"])
        qkv = (E @ p["qkv_w"] + p["qkv_b"]).reshape(N, T, 3, h, -1).permute(2, 0, 3, 1, 4)   # [3,N,h,T,dh]
        km = (ids != 0)
        bias = torch.zeros(N, 1, 1, T, dtype=E.dtype).masked_fill(~km[:, None, None, :], float("-inf"))
        S = qkv[0] @ qkv[1].transpose(-1, -2) / np.sqrt(qkv.shape[-1]) + bias
        allm = ~km.any(-1)
        S = torch.where(allm[:, None, None, None], torch.zeros_like(S), S)   # no NaN from an all-pad title
        P = torch.softmax(S, dim=-1) * km[:, None, None, :].to(E.dtype)
        c = (P @ qkv[2]).permute(0, 2, 1, 3).reshape(N, T, -1)
    else:
        k, D, Fo = p["conv_w"].shape
        pl = (k - 1) // 2
        x = F.pad(E.transpose(1, 2), (pl, k - 1 - pl))           # [N,D,T+k-1]
        w = p["conv_w"].permute(2, 1, 0)                        # [F,D,k]
        c = F.relu(F.conv1d(x, w, p["conv_b"])).transpose(1, 2)  # [N,T,F]
    c = c * (ids != 0).unsqueeze(-1).to(c.dtype)
    tmask = (c.detach() != 0).any(dim=-1)
    c = c * tmask.unsqueeze(-1).to(c.dtype)
    xx = c
    if keep2 is not None:
        xx = c * torch.as_tensor(keep2, dtype=c.dtype) * scale
    a = torch.tanh(xx @ p["att_w"] + p["att_b"])
    e = torch.exp(a) * tmask.to(a.dtype)
    wgt = e / (e.sum(dim=1, keepdim=True) + EPS)
    pooled = (xx * wgt.unsqueeze(-1)).sum(dim=1)
    return pooled @ p["dense_w"] + p["dense_b"]


def gru(p, x, mask, h0):
    B, W, _ = x.shape
    H = p["gru_u"].shape[0]
    h = h0
    for t in range(W):
        xt = x[:, t] @ p["gru_w"] + p["gru_b"]
        rec = h @ p["gru_u"][:, :2 * H]
        z = torch.clamp(0.2 * (xt[:, :H] + rec[:, :H]) + 0.5, 0, 1)
        r = torch.clamp(0.2 * (xt[:, H:2 * H] + rec[:, H:]) + 0.5, 0, 1)
        hh = torch.tanh(xt[:, 2 * H:] + (r * h) @ p["gru_u"][:, 2 * H:])
        hn = z * h + (1 - z) * hh
        h = torch.where(mask[:, t].unsqueeze(-1), hn, h)
    return h


def model_loss(p, arch, user, hist, cands, keep1=None, keep2=None, scale=1.0, keep_u=None, keep_u_scale=2.0):
    user = torch.as_tensor(np.asarray(user), dtype=torch.long).reshape(-1)
    hist_np = np.asarray(hist)
    B, W, T = hist_np.shape
    K1 = np.asarray(cands).shape[1]
    titles = np.concatenate([hist_np.reshape(B * W, T), np.asarray(cands).reshape(B * K1, T)], 0)
    v = news_encoder(p, titles, keep1, keep2, scale)
    hv = v[:B * W].reshape(B, W, -1)
    dv = v[B * W:].reshape(B, K1, -1)
    hmask = torch.as_tensor((hist_np != 0).any(-1))
    hv = hv * hmask.unsqueeze(-1).to(hv.dtype)
    gmask = (hv.detach() != 0).any(-1)
    uvec = p["user_emb"][user]
    uvec2 = p["user_emb2"][user] if arch == "iigru" else None
    if keep_u is not None:          # whole-row dropout of the user embedding: dgru (paper.py:610) / id_keep (cook.py:139-142)
        km = torch.as_tensor(keep_u, dtype=uvec.dtype).unsqueeze(-1) * keep_u_scale
        uvec = uvec * km
        if uvec2 is not None:
            uvec2 = uvec2 * km
    H = p["gru_u"].shape[0]
    if arch == "igru":
        u = gru(p, hv, gmask, uvec)
    elif arch == "nigru":
        u = gru(p, hv, gmask, torch.zeros(B, H, dtype=hv.dtype))
    elif arch == "pgru":
        u = gru(p, hv, gmask, torch.zeros(B, H, dtype=hv.dtype)) + uvec
    elif arch == "iigru":
        u = torch.cat([gru(p, hv, gmask, uvec), uvec2], dim=1) @ p["con_w"] + p["con_b"]
    elif arch == "vo":
        u = uvec
    elif arch == "niavg":
        u = hv.sum(1) / (gmask.to(hv.dtype).sum(1, keepdim=True) + EPS)
    elif arch == "niatt":
        a = torch.tanh(hv @ p["uatt_w"] + p["uatt_b"])
        e = torch.exp(a) * gmask.to(a.dtype)
        u = (hv * (e / (e.sum(1, keepdim=True) + EPS)).unsqueeze(-1)).sum(1)
    else:
        h = gru(p, hv, gmask, torch.zeros(B, H, dtype=hv.dtype))
        u = torch.cat([h, uvec], dim=1)
        if arch == "gru":
            u = u @ p["con_w"] + p["con_b"]
    if "sdu_w" in p:                                    # ddot (paper.py:452-455)
        z = (torch.tanh(u @ p["sdu_w"] + p["sdu_b"]).unsqueeze(1) * torch.tanh(dv @ p["sdd_w"] + p["sdd_b"])).sum(-1)
    elif "sn1_w" in p:                                  # dnn (paper.py:448-451)
        cat = torch.cat([u.unsqueeze(1).expand(-1, K1, -1), dv], dim=-1)
        z = torch.relu(cat @ p["sn1_w"] + p["sn1_b"]) @ p["sn2_w"] + p["sn2_b"]
    else:
        z = (u.unsqueeze(1) * dv).sum(-1)
    pr = torch.softmax(z, dim=-1)
    q = pr / pr.sum(-1, keepdim=True)
    q = torch.clamp(q, EPS, 1 - EPS)
    loss = -(torch.log(q[:, 0])).mean()
    return loss, z, pr


def user_vector(p, arch, user, hist, v):
    """user encoder on top of the doc vectors v ([B*W history rows | candidate rows])."""
    user = torch.as_tensor(np.asarray(user), dtype=torch.long).reshape(-1)
    hist_np = np.asarray(hist)
    B, W, T = hist_np.shape
    hv = v[:B * W].reshape(B, W, -1)
    hmask = torch.as_tensor((hist_np != 0).any(-1))
    hv = hv * hmask.unsqueeze(-1).to(hv.dtype)
    gmask = (hv.detach() != 0).any(-1)
    H = p["gru_u"].shape[0]
    if arch == "igru":
        return gru(p, hv, gmask, p["user_emb"][user])
    if arch == "iicgru":      # Seq2VecPaperId's `iigru` (paper.py:343-348): [igru | row of the second table], no Dense
        return torch.cat([gru(p, hv, gmask, p["user_emb"][user]), p["user_emb2"][user]], dim=1)
    h = gru(p, hv, gmask, torch.zeros(B, H, dtype=hv.dtype))
    if arch == "ngru":        # Seq2VecPaperId's `gru` (paper.py:339-342)
        return torch.cat([h, p["user_emb"][user]], dim=1)
    assert arch == "nigru"
    return h


def point_loss(p, arch, head, user, hist, cand, labels, gain, K, keep1=None, keep2=None, scale=1.0):
    """Pointwise sibling (paper.py:223-227, 258-262, 356-383; loss seq2vec.py:213-216), autograd form."""
    hist_np = np.asarray(hist)
    B, W, T = hist_np.shape
    titles = np.concatenate([hist_np.reshape(B * W, T), np.asarray(cand).reshape(B, T)], 0)
    v = news_encoder(p, titles, keep1, keep2, scale)
    u = user_vector(p, arch, user, hist, v)
    d = v[B * W:]
    if head == "point_dot":
        s = (u * d).sum(-1)
    else:
        hid = torch.relu(torch.cat([u, d], 1) @ p["score_w1"] + p["score_b1"])
        s = hid @ p["score_w2"] + p["score_b2"][0]
    pr = torch.sigmoid(s)
    y = torch.as_tensor(np.asarray(labels), dtype=pr.dtype).reshape(-1)
    loss = -0.5 * (1 + K) * torch.mean(y * torch.log(pr + 1e-8) * gain + (1 - y) * torch.log(1 - pr + 1e-8) / K)
    return loss, s, pr
