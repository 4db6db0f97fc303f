"""Diagnostic: bf16x3 conv kernels on random vs all-zero operands (DVFS check: cdna guide 5.4 rule 25/28) at the config-2
shapes.  python tools/diag_conv3.py"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mnexp_amd._lib import lib, check
from tools.bench_conv3 import stream, timeit

N, D, T, F, k = 3520, 300, 30, 400, 3
rng = np.random.default_rng(0)
ids = rng.integers(1, 1000, size=(N, T)).astype(np.int32)
lens = rng.integers(5, T + 6, size=N)
ids[np.arange(T)[None, :] >= lens[:, None]] = 0
ids_d = torch.from_numpy(ids).cuda()
vidx = torch.full((N * T,), -1, dtype=torch.int32, device='cuda')
vcount = torch.zeros(1, dtype=torch.int32, device='cuda')
vs = torch.zeros(max(1, lib.lstur_valid_tokens_scratch(N * T) // 4), dtype=torch.int32, device='cuda')
check(lib.lstur_valid_tokens(ids_d.data_ptr(), N * T, vidx.data_ptr(), vcount.data_ptr(), vs.data_ptr(), stream()))
cnt = int(vcount.item())
fl = 2.0 * cnt * k * D * F
for mode in ('random', 'zeros', 'random'):
    Ep = torch.zeros(N * (T + k - 1) * D + 256, device='cuda')
    W = torch.zeros(k * D, F, device='cuda')
    dC = torch.zeros(N * T, F, device='cuda')
    if mode == 'random':
        Ep[:N * (T + k - 1) * D] = torch.randn(N * (T + k - 1) * D, device='cuda') * 0.1
        W = torch.randn(k * D, F, device='cuda') * 0.05
        dC = torch.randn(N * T, F, device='cuda')
    b = torch.zeros(F, device='cuda')
    w3 = torch.zeros(lib.lstur_conv3_w3_elems(D, k, F), dtype=torch.int16, device='cuda')
    check(lib.lstur_conv3_pack_w(W.data_ptr(), D, k, F, w3.data_ptr(), stream()))
    out = torch.zeros(N * T, F, device='cuda')
    f = lambda: check(lib.lstur_conv3_relu_mask_fwd(Ep.data_ptr(), w3.data_ptr(), b.data_ptr(), ids_d.data_ptr(), out.data_ptr(),
                                                    N, T, D, k, F, vidx.data_ptr(), vcount.data_ptr(), stream()))
    us = timeit(f, 50)
    dw = torch.empty(k * D, F, device='cuda')
    scr = torch.empty(lib.lstur_conv3_wgrad_scratch(N, T, D, k, F), dtype=torch.uint8, device='cuda')
    g = lambda: check(lib.lstur_conv3_wgrad(Ep.data_ptr(), dC.data_ptr(), dw.data_ptr(), scr.data_ptr(), N, T, D, k, F,
                                            vidx.data_ptr(), vcount.data_ptr(), stream()))
    us2 = timeit(g, 50)
    print('%-7s fwd %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF (fp32-equivalent; x6 = bf16 executed)' % (
        mode, us, fl / us / 1e6, us2, fl / us2 / 1e6), flush=True)
