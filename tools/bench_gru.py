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
    packed = torch.zeros(lib.lstur_gru_pack_floats(H), device='cuda')
    check(lib.lstur_gru_pack_u(Um.data_ptr(), H, packed.data_ptr(), stream()))
    tape = torch.zeros(lib.lstur_gru_tape_floats(n, W, H), device='cuda')
    hlast = torch.empty(n, H, device='cuda')
    dg = torch.empty(n, W, 3 * H, device='cuda'); rh = torch.empty(n, W, H, device='cuda'); hp = torch.empty(n, W, H, device='cuda')
    dh0 = torch.empty(n, H, device='cuda'); dh = torch.randn(n, H, device='cuda', generator=g)
    def fwd():
        check(lib.lstur_gru_fwd(xw.data_ptr(), gm.data_ptr(), h0.data_ptr(), packed.data_ptr(), tape.data_ptr(),
                                hlast.data_ptr(), n, W, H, stream()))
    def bwd():
        check(lib.lstur_gru_bwd(dh.data_ptr(), gm.data_ptr(), tape.data_ptr(), packed.data_ptr(), dg.data_ptr(),
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
    if int(os.environ.get('LSTUR_GRU4_DEBUG', '0')) & 8:
        hlast.zero_(); fwd(); torch.cuda.synchronize()
        v = hlast.flatten()[:64].cpu().numpy().reshape(8, 8)[:, :6] / W
        print('  per-step cycles per wave: mfma1, epi1, bar1, mfma2, epi2, bar2'); print(v.round(0)); print('  sum', v.sum(1).round(0))
    return res

if __name__ == '__main__':
    cfgs = [(64, 50, 200), (64, 100, 200)] if len(sys.argv) > 1 else [(16, 50, 200), (64, 50, 200), (64, 100, 200), (512, 50, 200), (64, 50, 128)]
    out = {}
    for (n, W, H) in cfgs:
        r = run(n, W, H)
        out[(n, W, H)] = r
        print('n=%d W=%d H=%d  fwd %.1f us (%.2f us/step)  bwd %.1f us (%.2f us/step)  dbg=%s generic=%s' % (
            n, W, H, r['fwd'], r['fwd'] / W, r['bwd'], r['bwd'] / W, os.environ.get('LSTUR_GRU4_DEBUG'),
            os.environ.get('LSTUR_GRU_GENERIC')))
    if (64, 50, 200) in out and (64, 100, 200) in out:
        a, b = out[(64, 50, 200)], out[(64, 100, 200)]
        print('marginal step: fwd %.2f us  bwd %.2f us ; fixed: fwd %.1f us bwd %.1f us' % (
            (b['fwd'] - a['fwd']) / 50, (b['bwd'] - a['bwd']) / 50, 2 * a['fwd'] - b['fwd'], 2 * a['bwd'] - b['bwd']))
