"""micro-benchmark: the dense GEMMs of the step's critical path, bf16x3 kernel vs exact-fp32 MFMA kernel (back-to-back launches)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mnexp_amd._lib import lib, check

def stream():
    return torch.cuda.current_stream().cuda_stream

shapes = [('dense fwd', 3520, 200, 400, 0), ('xw', 3200, 600, 200, 0), ('dx', 3200, 200, 600, 1), ('dpooled', 3520, 400, 200, 1)]
g = torch.Generator(device='cuda').manual_seed(0)
for name, M, N, K, b_nk in shapes:
    A = torch.randn(M, K, device='cuda', generator=g)
    B = torch.randn((N, K) if b_nk else (K, N), device='cuda', generator=g)
    bias = torch.randn(N, device='cuda', generator=g)
    C = torch.empty(M, N, device='cuda')
    scratch = torch.empty(max(1, lib.lstur_gemm_scratch(M, N, K)), dtype=torch.uint8, device='cuda')
    def f3():
        check(lib.lstur_gemm_bf16x3(A.data_ptr(), K, B.data_ptr(), B.stride(0), b_nk, C.data_ptr(), N, bias.data_ptr(), M, N, K, stream()))
    def f32():
        check(lib.lstur_gemm_f32(A.data_ptr(), K, 0, B.data_ptr(), B.stride(0), b_nk, C.data_ptr(), N, bias.data_ptr(), M, N, K, 1.0, scratch.data_ptr(), stream()))
    out = []
    for f in (f3, f32):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            f()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 10)
    print('%-10s M=%d N=%d K=%d  bf16x3 %.1f us   fp32 %.1f us' % (name, M, N, K, out[0], out[1]))
