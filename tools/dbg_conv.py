import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mnexp_amd._lib import lib, check
from test_gpu_kernels import dev, stream, pack_b
rng = np.random.default_rng(1)
for (N, D, T, F, k) in [(3, 300, 30, 400, 3), (3, 304, 30, 400, 3), (3, 48, 30, 400, 3), (3, 300, 30, 80, 3)]:
    E = rng.normal(0, 0.3, size=(N, T, D)).astype(np.float32)
    ids = np.ones((N, T), np.int32)
    Wc = rng.normal(0, 0.05, size=(k, D, F)).astype(np.float32)
    b = np.zeros(F, np.float32)
    Ep = np.zeros((N, T + 2, D), np.float32); Ep[:, 1:-1] = E
    Epd = torch.zeros(N * (T + 2) * D + 256, dtype=torch.float32, device='cuda'); Epd[:N * (T + 2) * D] = dev(Ep).view(-1)
    Wd = dev(Wc.reshape(k * D, F)); Wp = pack_b(Wd)
    out = torch.empty(N * T, F, dtype=torch.float32, device='cuda')
    bd, idd = dev(b), dev(ids)
    check(lib.lstur_conv1d_relu_mask_fwd(Epd.data_ptr(), Wp.data_ptr(), bd.data_ptr(), idd.data_ptr(), out.data_ptr(), N, T, D, k, F, stream()))
    x = torch.nn.functional.pad(dev(E).double().transpose(1, 2), (1, 1))
    ref = torch.relu(torch.nn.functional.conv1d(x, dev(Wc).double().permute(2, 1, 0), dev(b).double())).transpose(1, 2).reshape(N*T, F)
    err = (out.double() - ref).abs()
    print((N, D, T, F, k), 'max err', err.max().item())
    bad = (err > 1e-4)
    print(' bad rows:', bad.any(1).nonzero().flatten().tolist()[:40])
    print(' bad cols:', bad.any(0).nonzero().flatten().tolist()[:40])
