"""Micro-benchmark of the split-operand conv family (convx.hip) at the config-2 shapes on realistic token lists (62 % non-pad),
interleaved rounds in one process (cdna guide 5.4 rule 24): gather, forward and kernel gradient for parts = 2 (f16x2) and
3 (bf16x3), hot (back to back) and from cold caches.  LSTUR_LIB=mnexp_amd/liblstur_hip_ablate.so + --ablate adds the
timing-only ablation variants (bench-only library, `make -C mnexp_amd/csrc ablate`).
    python tools/bench_convx.py [n_titles] [rounds] [--F=400]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mnexp_amd._lib import lib, check  # noqa: E402


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


def time_hot(f, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    N = int(args[0]) if len(args) > 0 else 3520
    rounds = int(args[1]) if len(args) > 1 else 6
    D, T, F, k, V = 300, 30, 400, 3, 50000
    for a in sys.argv[1:]:
        if a.startswith('--F='):
            F = int(a[4:])
    rng = np.random.default_rng(0)
    ids = rng.integers(1, V, size=(N, T)).astype(np.int32)
    lens = np.clip(rng.integers(5, T + 6, size=N), 0, T)
    lens[rng.random(N) < 0.14] = 0                      # front-padded history slots: all-pad titles
    ids[np.arange(T)[None, :] >= lens[:, None]] = 0
    ids_d = torch.from_numpy(ids).cuda()
    live = torch.from_numpy(((ids != 0).astype(np.uint64) << np.arange(T, dtype=np.uint64)).sum(1).astype(np.uint32).view(np.int32)).cuda()
    table = torch.randn(V, D, device='cuda') * 0.1
    table[0] = 0
    W = torch.randn(k * D, F, device='cuda') * 0.05
    b = torch.randn(F, device='cuda') * 0.1
    dC = torch.randn(N * T, F, device='cuda')
    out = torch.empty(N * T, F, device='cuda')
    dw = torch.empty(k * D, F, device='cuda')
    vidx = torch.empty(N * T, dtype=torch.int32, device='cuda')
    vcount = torch.zeros(1, dtype=torch.int32, device='cuda')
    vs = torch.zeros(max(1, lib.lstur_valid_tokens_scratch(N * T) // 4), dtype=torch.int32, device='cuda')
    check(lib.lstur_valid_tokens(ids_d.data_ptr(), N * T, vidx.data_ptr(), vcount.data_ptr(), vs.data_ptr(), stream()))
    torch.cuda.synchronize()
    cnt = int(vcount.item())
    nonpad = ids != 0
    q = np.pad(nonpad, [(0, 0), (2, 2)])
    rows = int((q[:, 0:T + 2] | q[:, 1:T + 3] | q[:, 2:T + 4]).sum())
    Dp = (D + 7) // 8 * 8
    print('%d titles, %d tokens, %d non-pad (%.3f), %d embedding rows covered' % (N, N * T, cnt, cnt / (N * T), rows))
    arms = []
    for parts in (2, 3):
        cs = torch.zeros(lib.lstur_convx_scales_bytes() // 4, dtype=torch.float32, device='cuda')
        check(lib.lstur_convx_scales_init(cs.data_ptr(), stream()))
        check(lib.lstur_convx_embedding_scale(table.data_ptr(), V * D, 0.2, cs.data_ptr(), stream()))
        ex = torch.zeros(lib.lstur_convx_e_elems(N, T, D, k, parts), dtype=torch.int16, device='cuda')
        wx = torch.zeros(lib.lstur_convx_w_elems(D, k, F, parts), dtype=torch.int16, device='cuda')
        check(lib.lstur_convx_pack_w(W.data_ptr(), D, k, F, wx.data_ptr(), parts, cs.data_ptr(), stream()))
        scr = torch.empty(lib.lstur_convx_fwd_scratch(), dtype=torch.uint8, device='cuda')
        wscr = torch.empty(lib.lstur_convx_wgrad_scratch(N, T, D, k, F), dtype=torch.uint8, device='cuda')
        offs = torch.zeros(2 * N * T, dtype=torch.int32, device='cuda')
        keep = (cs, ex, wx, scr, wscr, offs)
        g = lambda parts=parts, cs=cs, ex=ex: check(lib.lstur_convx_gather(
            table.data_ptr(), ids_d.data_ptr(), ex.data_ptr(), N, T, D, 1, 1, 0.2, 7, 1, parts, cs.data_ptr(), live.data_ptr(), stream()))
        f = lambda parts=parts, cs=cs, ex=ex, wx=wx, scr=scr: check(lib.lstur_convx_fwd(
            ex.data_ptr(), wx.data_ptr(), b.data_ptr(), ids_d.data_ptr(), out.data_ptr(), N, T, D, k, F, vidx.data_ptr(),
            vcount.data_ptr(), parts, cs.data_ptr(), scr.data_ptr(), stream()))
        w = lambda parts=parts, cs=cs, ex=ex, wscr=wscr, offs=offs: check(lib.lstur_convx_wgrad(
            ex.data_ptr(), dC.data_ptr(), dw.data_ptr(), wscr.data_ptr(), offs.data_ptr(), N, T, D, k, F, vidx.data_ptr(),
            vcount.data_ptr(), parts, cs.data_ptr(), stream()))
        g()
        moved = rows * (4 * D + 2 * parts * Dp) + N * T * 4
        fl = 2.0 * cnt * k * D * F
        arms += [('gather  parts=%d' % parts, g, 'GB/s', moved / 1e3, keep),
                 ('forward parts=%d' % parts, f, 'TFLOP/s', fl / 1e6, keep),
                 ('wgrad   parts=%d (+absmax, offsets, reduce)' % parts, w, 'TFLOP/s', fl / 1e6, keep)]
    if '--ablate' in sys.argv:      # bench-only library (LSTUR_LIB=.../liblstur_hip_ablate.so): WRONG results, timing only
        two = [a for a in arms if 'parts=2' in a[0] and not a[0].startswith('gather')]
        def with_env(fn, key, v):
            def run():
                os.environ[key] = str(v)
                fn()
                os.environ.pop(key)
            return run
        names = {10: 'no global loads in the loop', 11: '+ no LDS stores', 12: '+ no barrier', 13: '+ no B fragment reads',
                 14: 'only: no barrier', 15: 'only: no B fragment reads'}
        for v, nm in names.items():
            arms.append(('forward parts=2 ablate %d (%s)' % (v, nm), with_env(two[0][1], 'LSTUR_CONVX_FWD_VARIANT', v), 'TFLOP/s', two[0][3], None))
        for v, nm in {1: 'row offsets of step 0 only', 4: 'no global loads in the loop', 7: '1 + split skipped + 4', 15: '+ no LDS stores'}.items():
            arms.append(('wgrad parts=2 ablate %d (%s)' % (v, nm), with_env(two[1][1], 'LSTUR_CONVX_WGRAD_VARIANT', v), 'TFLOP/s', two[1][3], None))
    res = {n: [[], []] for n, *_ in arms}
    for _ in range(rounds):
        for n, fn, *_ in arms:
            res[n][0].append(time_hot(fn))
            res[n][1].append(time_cold(fn))
    for n, fn, unit, work, _ in arms:
        hot, cold = float(np.median(res[n][0])), float(np.median(res[n][1]))
        print('%-46s hot %7.1f us (%7.1f %s)   cold %7.1f us (%7.1f %s)' % (n, hot, work / hot, unit, cold, work / cold, unit))


if __name__ == '__main__':
    main()
