"""Micro-benchmark of the bf16x3 conv kernels (lstur_conv3_relu_mask_fwd / lstur_conv3_wgrad) at the config-2 shapes over
a compacted token list (62 % non-pad), HIP events; for kernel tuning.   python tools/bench_conv3.py [n_titles]"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mnexp_amd._lib import lib, check


def stream():
    return torch.cuda.current_stream().cuda_stream


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == '__main__':
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3520
    D, T, F, k = 300, 30, 400, 3
    rng = np.random.default_rng(0)
    ids = rng.integers(1, 1000, size=(N, T)).astype(np.int32)
    lens = rng.integers(5, T + 6, size=N)
    ids[np.arange(T)[None, :] >= lens[:, None]] = 0
    ids_d = torch.from_numpy(ids).cuda()
    Ep = torch.zeros(N * (T + k - 1) * D + 256, device='cuda')
    Ep[:N * (T + k - 1) * D] = torch.randn(N * (T + k - 1) * D, device='cuda') * 0.1
    W = torch.randn(k * D, F, device='cuda') * 0.05
    b = torch.randn(F, device='cuda') * 0.1
    w3 = torch.zeros(lib.lstur_conv3_w3_elems(D, k, F), dtype=torch.int16, device='cuda')
    check(lib.lstur_conv3_pack_w(W.data_ptr(), D, k, F, w3.data_ptr(), stream()))
    vidx = torch.full((N * T,), -1, dtype=torch.int32, device='cuda')
    vcount = torch.zeros(1, dtype=torch.int32, device='cuda')
    vs = torch.zeros(max(1, lib.lstur_valid_tokens_scratch(N * T) // 4), dtype=torch.int32, device='cuda')
    check(lib.lstur_valid_tokens(ids_d.data_ptr(), N * T, vidx.data_ptr(), vcount.data_ptr(), vs.data_ptr(), stream()))
    cnt = int(vcount.item())
    out = torch.zeros(N * T, F, device='cuda')
    f = lambda: check(lib.lstur_conv3_relu_mask_fwd(Ep.data_ptr(), w3.data_ptr(), b.data_ptr(), ids_d.data_ptr(), out.data_ptr(),
                                                    N, T, D, k, F, vidx.data_ptr(), vcount.data_ptr(), stream()))
    us = timeit(f)
    fl = 2.0 * cnt * k * D * F
    print('conv3 fwd   %d tokens (%d rows)  %7.1f us  %6.1f TFLOP/s fp32-equivalent' % (N * T, cnt, us, fl / us / 1e6))
    dC = torch.randn(N * T, F, device='cuda')
    dw = torch.empty(k * D, F, device='cuda')
    scr = torch.empty(lib.lstur_conv3_wgrad_scratch(N, T, D, k, F), dtype=torch.uint8, device='cuda')
    f = lambda: check(lib.lstur_conv3_wgrad(Ep.data_ptr(), dC.data_ptr(), dw.data_ptr(), scr.data_ptr(), N, T, D, k, F,
                                            vidx.data_ptr(), vcount.data_ptr(), stream()))
    us = timeit(f)
    print('conv3 wgrad %d tokens (%d rows)  %7.1f us  %6.1f TFLOP/s fp32-equivalent (incl. slab reduction)' % (
        N * T, cnt, us, fl / us / 1e6))
