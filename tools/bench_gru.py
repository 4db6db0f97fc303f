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
    sixth = lib.lstur_gru_pack_floats(H) // 6
    uzr, uh, uth, utzr = (torch.zeros(k * sixth, device='cuda') for k in (2, 1, 1, 2))
    check(lib.lstur_gru_pack_u(Um.data_ptr(), H, uzr.data_ptr(), uh.data_ptr(), uth.data_ptr(), utzr.data_ptr(), stream()))
    tape = torch.zeros(lib.lstur_gru_tape_floats(n, W, H), device='cuda')
    hlast = torch.empty(n, H, device='cuda')
    dg = torch.empty(n, W, 3 * H, device='cuda'); rh = torch.empty(n, W, H, device='cuda'); hp = torch.empty(n, W, H, device='cuda')
    dh0 = torch.empty(n, H, device='cuda'); dh = torch.randn(n, H, device='cuda', generator=g)
    def fwd():
        check(lib.lstur_gru_fwd(xw.data_ptr(), gm.data_ptr(), h0.data_ptr(), uzr.data_ptr(), uh.data_ptr(), tape.data_ptr(),
                                hlast.data_ptr(), n, W, H, stream()))
    def bwd():
        check(lib.lstur_gru_bwd(dh.data_ptr(), gm.data_ptr(), tape.data_ptr(), utzr.data_ptr(), uth.data_ptr(), dg.data_ptr(),
                                rh.data_ptr(), hp.data_ptr(), dh0.data_ptr(), n, W, H, stream()))
    res = {}
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters * 1e3
    if int(os.environ.get('LSTUR_GRU_DEBUG', '0')) & 16:
        fwd(); torch.cuda.synchronize(); S = tape.numel() // 11; cyc, ticks = tape[8 * S].item(), tape[8 * S + 1].item()
        print('  in-kernel: %.0f shader cycles over %.1f us -> %.3f GHz' % (cyc, ticks / 100.0, cyc / (ticks / 100.0) / 1e3))
    if int(os.environ.get('LSTUR_GRU_DEBUG', '0')) & 32:
        fwd(); torch.cuda.synchronize(); S = tape.numel() // 11; v = tape[7 * S:7 * S + 64].cpu().numpy().reshape(8, 8)[:, :6] / W
        print('  per-step cycles per wave: gemm1, epi1, bar1, gemm2, epi2, bar2'); print(v.round(0))
    return res

if __name__ == '__main__':
    cfgs = [(64, 50, 200)] if len(sys.argv) > 1 else [(16, 50, 200), (64, 50, 200), (512, 50, 200), (64, 50, 128), (64, 50, 256)]
    for (n, W, H) in cfgs:
        r = run(n, W, H)
        print('n=%d W=%d H=%d  fwd %.1f us (%.2f us/step)  bwd %.1f us (%.2f us/step)  nostream=%s' % (
            n, W, H, r['fwd'], r['fwd'] / W, r['bwd'], r['bwd'] / W, os.environ.get('LSTUR_GRU_DEBUG')))
