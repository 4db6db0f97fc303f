"""Micro-benchmark + check of the bf16 projection GEMMs of the mhsa encoder (lstur_proj_fwd_bf16 / _wgrad_bf16)
at the config-2 token count (HIP events); for kernel tuning.   python tools/bench_proj.py [M] [D] [N]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mnexp_amd._lib import lib, check


def stream():
    return torch.cuda.current_stream().cuda_stream


def timeit(f, iters=30):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 105600
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    g = torch.Generator(device='cuda').manual_seed(0)
    E = torch.randn(M, D, device='cuda', generator=g) * 0.1
    W = torch.randn(D, N, device='cuda', generator=g) * 0.1
    b = torch.randn(N, device='cuda', generator=g) * 0.1
    out = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    f = lambda: check(lib.lstur_proj_fwd_bf16(E.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, D, stream()))
    us = timeit(f)
    ref = (E.bfloat16().float() @ W.bfloat16().float() + b)
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    print('proj fwd   %6dx%4dx%4d  %7.1f us  %6.1f TFLOP/s  rel err %.2e' % (M, N, D, us, 2.0 * M * N * D / us / 1e6, err))
    G = (torch.randn(M, N, device='cuda', generator=g) * 0.1).bfloat16()
    dw = torch.empty(D + 1, N, device='cuda')
    scratch = torch.empty(lib.lstur_proj_wgrad_scratch(D, N, M), dtype=torch.uint8, device='cuda')
    f = lambda: check(lib.lstur_proj_wgrad_bf16(E.data_ptr(), G.data_ptr(), dw.data_ptr(), scratch.data_ptr(), D, N, M, stream()))
    us = timeit(f)
    ref = torch.cat([E.bfloat16().float().t() @ G.float(), G.float().sum(0, keepdim=True)], 0)
    err = (dw - ref).abs().max().item() / ref.abs().max().item()
    print('proj wgrad %6dx%4dx%4d  %7.1f us  %6.1f TFLOP/s  rel err %.2e  (incl. slab reduction)' % (
        D, N, M, us, 2.0 * M * N * D / us / 1e6, err))
