"""which gradient slots differ between the fused step and the split calls (diagnostics)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_gpu_model import MID, build
from test_gpu_train_step import _batches, _state
from mnexp_amd._lib import lib, check
B, W, T, K1 = 24, 7, 11, 5
outs = []
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for mode in ('fused', 'split', 'fused'):
    O, p, model, rng = build('igru', MID, B, W, T, K1, 0.2)
    steps = _batches(model, B, W, K1, T, nsteps, MID['N_u'])
    for i, (u, h, c) in enumerate(steps):
        if mode == 'fused':
            model._train_step(u, h, c, True)
        else:
            s = model._stream()
            check(lib.lstur_forward_backward_ids(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), B, i, 1.0, model.stats.data_ptr(), s), 'fb')
            check(lib.lstur_adam_step(model._ctx, 1e-3, i + 1, None, None, 0, s), 'adam')
    torch.cuda.synchronize()
    outs.append((_state(model), model.layout))
for a, b, tag in ((0, 1, 'fused vs split'), (0, 2, 'fused vs fused')):
    for k in outs[a][0]:
        x, y = outs[a][0][k], outs[b][0][k]
        if not np.array_equal(x, y):
            print(tag, k, 'differs at', np.flatnonzero(x != y)[:10], 'count', (x != y).sum(), 'max', np.abs(x - y).max())
            if k.startswith('dense'):
                for n, off, cnt in outs[0][1]:
                    d = (x[off:off + cnt] != y[off:off + cnt]).sum()
                    if d: print('   ', n, d, 'of', cnt)
print('done')
