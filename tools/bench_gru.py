"""Micro-benchmark of lstur_gru_fwd / lstur_gru_bwd alone (HIP events), for kernel tuning."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from mnexp_amd._lib import lib, check

def stream():
    return torch.cuda.current_stream().cuda_stream

def run(n, W, H, iters=20):
    U = H
    g = torch.Generator(device='cuda').manual_seed(0)
    xw = torch.randn(n, W, 3 * H, device='cuda', generator=g) * 0.5
    gm = torch.ones(n, W, dtype=torch.uint8, device='cuda')
    h0 = torch.randn(n, H, device='cuda', generator=g) * 0.05
    Um = torch.randn(H, 3 * H, device='cuda', generator=g) * 0.07
    def pack(B, trans=False, K=None, N=None):
        if not trans:
            K, N = B.shape
            out = torch.zeros(lib.lstur_pack_b_floats(K, N), device='cuda')
            check(lib.lstur_pack_b(B.data_ptr(), B.stride(0), K, N, out.data_ptr(), stream()))
        else:
            out = torch.zeros(lib.lstur_pack_b_floats(K, N), device='cuda')
            check(lib.lstur_pack_bt(B.data_ptr(), 3 * H, K, N, out.data_ptr(), stream()))
        return out
    uzr, uh = pack(Um[:, :2 * H]), pack(Um[:, 2 * H:])
    utzr, uth = pack(Um, True, 2 * H, H), pack(Um[:, 2 * H:], True, H, H)
    hs = torch.empty(n, W + 1, H, device='cuda'); zs, rs, hhs = (torch.empty(n, W, H, device='cuda') for _ in range(3))
    dg = torch.empty(n, W, 3 * H, device='cuda'); rh = torch.empty(n, W, H, device='cuda'); hp = torch.empty(n, W, H, device='cuda')
    dh0 = torch.empty(n, H, device='cuda'); dh = torch.randn(n, H, device='cuda', generator=g)
    def fwd():
        check(lib.lstur_gru_fwd(xw.data_ptr(), gm.data_ptr(), h0.data_ptr(), uzr.data_ptr(), uh.data_ptr(), hs.data_ptr(),
                                zs.data_ptr(), rs.data_ptr(), hhs.data_ptr(), n, W, H, stream()))
    def bwd():
        check(lib.lstur_gru_bwd(dh.data_ptr(), gm.data_ptr(), hs.data_ptr(), zs.data_ptr(), rs.data_ptr(), hhs.data_ptr(),
                                utzr.data_ptr(), uth.data_ptr(), dg.data_ptr(), rh.data_ptr(), hp.data_ptr(), dh0.data_ptr(), n, W, H, stream()))
    res = {}
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters * 1e3
    return res

if __name__ == '__main__':
    for (n, W, H) in [(16, 50, 200), (64, 50, 200), (512, 50, 200), (64, 50, 128), (64, 50, 256)]:
        r = run(n, W, H)
        print('n=%d W=%d H=%d  fwd %.1f us (%.2f us/step)  bwd %.1f us (%.2f us/step)  nostream=%s' % (
            n, W, H, r['fwd'], r['fwd'] / W, r['bwd'], r['bwd'] / W, os.environ.get('LSTUR_GRU_NOSTREAM')))
