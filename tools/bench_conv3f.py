"""A/B of the bf16x3 conv forward kernels at the config-2 shapes on realistic token lists (62 % non-pad), interleaved rounds
in one process (cdna guide 5.4 rule 24): v1 (conv3.hip, fp32 E split in the loader) against conv3f.hip on pre-split
operands, variants 0 / 1, + the gather in both forms.   python tools/bench_conv3f.py [n_titles] [rounds]"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mnexp_amd._lib import lib, check


def stream():
    return torch.cuda.current_stream().cuda_stream


_flush = None


def time_cold(f, iters=6):
    """Mean time of f() from COLD caches: 600 MB are rewritten before every call (the Infinity Cache holds 256 MiB), as
    between two uses of a buffer inside a training step."""
    global _flush
    if _flush is None:
        _flush = torch.empty(150 * 1024 * 1024, dtype=torch.float32, device='cuda')
    tot = 0.0
    for _ in range(iters):
        _flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


def time_once(f, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    N = int(args[0]) if len(args) > 0 else 3520
    rounds = int(args[1]) if len(args) > 1 else 8
    frac = float(args[2]) if len(args) > 2 else 0.14
    D, T, F, k, V = 300, 30, 400, 3, 50000
    for a in sys.argv[1:]:
        if a.startswith('--F='):
            F = int(a[4:])       # (column-tile experiments)
    rng = np.random.default_rng(0)
    ids = rng.integers(1, V, size=(N, T)).astype(np.int32)
    lens = np.clip(rng.integers(5, T + 6, size=N), 0, T)
    lens[rng.random(N) < frac] = 0                      # front-padded history slots: all-pad titles
    ids[np.arange(T)[None, :] >= lens[:, None]] = 0
    ids_d = torch.from_numpy(ids).cuda()
    table = torch.randn(V, D, device='cuda') * 0.1
    table[0] = 0
    W = torch.randn(k * D, F, device='cuda') * 0.05
    b = torch.randn(F, device='cuda') * 0.1
    Tp = T + k - 1
    Ep = torch.zeros(N * Tp * D + 256, device='cuda')
    e3 = torch.zeros(lib.lstur_conv3_e3_elems(N, T, D, k), dtype=torch.int16, device='cuda')
    w3 = torch.zeros(lib.lstur_conv3_w3_elems(D, k, F), dtype=torch.int16, device='cuda')
    w3b = torch.zeros(lib.lstur_conv3_w3_elems2(D, k, F), dtype=torch.int16, device='cuda')
    check(lib.lstur_conv3_pack_w(W.data_ptr(), D, k, F, w3.data_ptr(), stream()))
    check(lib.lstur_conv3_pack_w2(W.data_ptr(), D, k, F, w3b.data_ptr(), stream()))
    vidx = torch.full((N * T,), -1, dtype=torch.int32, device='cuda')
    vcount = torch.zeros(1, dtype=torch.int32, device='cuda')
    vs = torch.zeros(max(1, lib.lstur_valid_tokens_scratch(N * T) // 4), dtype=torch.int32, device='cuda')
    check(lib.lstur_valid_tokens(ids_d.data_ptr(), N * T, vidx.data_ptr(), vcount.data_ptr(), vs.data_ptr(), stream()))
    cnt = int(vcount.item())
    out = torch.zeros(N * T, F, device='cuda')
    out2 = torch.zeros(N * T, F, device='cuda')
    g1 = lambda: check(lib.lstur_gather_f32(table.data_ptr(), ids_d.data_ptr(), Ep.data_ptr(), N, T, D, 1, 1, 0.2, 7, stream()))
    g3 = lambda: check(lib.lstur_gather_split3(table.data_ptr(), ids_d.data_ptr(), e3.data_ptr(), N, T, D, 1, 1, 0.2, 7, 0, stream()))
    g3s = lambda: check(lib.lstur_gather_split3(table.data_ptr(), ids_d.data_ptr(), e3.data_ptr(), N, T, D, 1, 1, 0.2, 7, 1, stream()))
    f1 = lambda: check(lib.lstur_conv3_relu_mask_fwd(Ep.data_ptr(), w3.data_ptr(), b.data_ptr(), ids_d.data_ptr(), out.data_ptr(),
                                                     N, T, D, k, F, vidx.data_ptr(), vcount.data_ptr(), stream()))
    scr = torch.empty(lib.lstur_conv3_fwd2_scratch(), dtype=torch.uint8, device='cuda')
    f2 = lambda v, sl=True: (lambda: check(lib.lstur_conv3_relu_mask_fwd2(
        e3.data_ptr(), w3b.data_ptr(), b.data_ptr(), ids_d.data_ptr(), out2.data_ptr(), N, T, D, k, F, vidx.data_ptr(),
        vcount.data_ptr(), v, scr.data_ptr() if sl else None, stream())))
    dC = torch.randn(N * T, F, device='cuda')
    dw = torch.empty(k * D, F, device='cuda')
    dw2 = torch.empty(k * D, F, device='cuda')
    ws1 = torch.empty(lib.lstur_conv3_wgrad_scratch(N, T, D, k, F), dtype=torch.uint8, device='cuda')
    ws2 = torch.empty(lib.lstur_conv3_wgrad2_scratch(N, T, D, k, F), dtype=torch.uint8, device='cuda')
    offs = torch.zeros(2 * N * T, dtype=torch.int32, device='cuda')
    w1 = lambda: check(lib.lstur_conv3_wgrad(Ep.data_ptr(), dC.data_ptr(), dw.data_ptr(), ws1.data_ptr(), N, T, D, k, F,
                                             vidx.data_ptr(), vcount.data_ptr(), stream()))
    w2 = lambda: check(lib.lstur_conv3_wgrad2(e3.data_ptr(), dC.data_ptr(), dw2.data_ptr(), ws2.data_ptr(), offs.data_ptr(), N, T, D,
                                              k, F, vidx.data_ptr(), vcount.data_ptr(), stream()))
    g1(); g3(); f1(); f2(0)(); w1(); w2()
    torch.cuda.synchronize()
    print('wgrad max |v2 - v1| = %.3g (scale %.3g)' % ((dw - dw2).abs().max().item(), dw.abs().max().item()))
    torch.cuda.synchronize()
    print('max |v2 - v1| = %.3g (scale %.3g)' % ((out - out2).abs().max().item(), out.abs().max().item()))
    arms = [('gather_f32', g1), ('gather_split3', g3), ('gather_split3 skip', g3s), ('conv3 v1', f1), ('conv3f', f2(0)), ('conv3f no-tail', f2(0, False)),
            ('wgrad v1 (+offsets, reduce)', w1), ('wgrad v2 (+offsets, reduce)', w2)]
    if '--ablate' in sys.argv:
        arms += [('conv3f abl%d' % v, f2(v, False)) for v in range(10, 17)]     # 16: MFMA shape probe (16x16x32)
    res = {n: [] for n, _ in arms}
    for f in [a[1] for a in arms]:
        for _ in range(3):
            f()
    cold = '--cold' in sys.argv
    for r in range(rounds):
        for n, f in arms:
            res[n].append(time_cold(f) if cold else time_once(f))
    fl = 2.0 * cnt * k * D * F
    print('%d titles, %d tokens, %d non-pad rows (%.1f %%)' % (N, N * T, cnt, 100.0 * cnt / (N * T)))
    for n, _ in arms:
        v = np.array(res[n])
        extra = ''
        if n.startswith('conv') or n.startswith('wgrad'):
            extra = '  %6.1f TFLOP/s fp32-equivalent, %.3f of the bf16x3 roofline' % (fl / np.median(v) / 1e6, fl / np.median(v) / 1e6 / 416.7)
        print('%-14s median %7.1f us  min %7.1f us%s' % (n, np.median(v), v.min(), extra), flush=True)
