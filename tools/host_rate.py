"""How long does the HOST take to enqueue one training step (no synchronisation inside the loop), against the GPU time per
step?  python tools/host_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from mnexp_amd.native_model import NativeModel

if __name__ == '__main__':
    wl = bench.WORKLOADS['lstur_ini_b64']
    titles, word, _ = bench.synth_inputs(wl, np.random.default_rng(1234), 0)
    _, _, batches = bench.synth_inputs(wl, np.random.default_rng(1234), 32)
    model = NativeModel(word, wl['N_u'], wl['B'], wl['W'], wl['T'], wl['K1'] - 1, wl['F'], wl['k'], wl['U'],
                        arch=wl['arch'], dropout=0.2, lr=1e-3, seed=7, device='cuda:0', title_table=titles)
    dev = [(model._dev_i32(u), model._dev_i32(h), model._dev_i32(c)) for u, h, c in batches]
    for k in range(10):
        model._train_step(*dev[k % len(dev)], True)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for k in range(n):
        model._train_step(*dev[k % len(dev)], True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host enqueue %.1f us/step; wall incl. drain %.1f us/step' % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
