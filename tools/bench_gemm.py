"""Micro-benchmark of lstur_gemm_f32 on the LSTUR step's GEMM shapes (HIP events), for kernel tuning."""
import sys, os, torch
sys.path.insert(0, '.')
from mnexp_amd._lib import lib, check

def stream():
    return torch.cuda.current_stream().cuda_stream

SHAPES = [  # ta, tb, M, N, K, name
    (0, 1, 3520, 200, 400, 'dense fwd'), (0, 1, 3200, 600, 200, 'gru xw'), (0, 1, 3200, 200, 600, 'd docvec'),
    (0, 1, 3520, 400, 200, 'd pooled'), (1, 0, 200, 600, 3200, 'd gru_w'), (1, 0, 200, 400, 3200, 'd gru_u zr'),
    (1, 0, 200, 200, 3200, 'd gru_u h'), (1, 0, 400, 200, 3520, 'd dense_w')]

def run(ta, tb, M, N, K, iters=50):
    g = torch.Generator(device='cuda').manual_seed(0)
    A = torch.randn((K, M) if ta else (M, K), device='cuda', generator=g)
    B = torch.randn((N, K) if tb else (K, N), device='cuda', generator=g)
    Cm = torch.empty(M, N, device='cuda')
    scratch = torch.empty(max(1, lib.lstur_gemm_scratch(M, N, K)), dtype=torch.uint8, device='cuda')
    def f():
        check(lib.lstur_gemm_f32(A.data_ptr(), A.stride(0), ta, B.data_ptr(), B.stride(0), tb, Cm.data_ptr(), N, None,
                                 M, N, K, 1.0, scratch.data_ptr(), stream()))
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

if __name__ == '__main__':
    tot = 0
    for ta, tb, M, N, K, name in SHAPES:
        us = run(ta, tb, M, N, K)
        tot += us
        print('%-12s ta=%d tb=%d %5dx%4dx%5d  %6.1f us  %6.1f TFLOP/s' % (name, ta, tb, M, N, K, us, 2.0 * M * N * K / us / 1e6))
    print('total %.1f us  (LSTUR_GEMM_TILED=%s)' % (tot, os.environ.get('LSTUR_GEMM_TILED')))
