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
    # + prefetch announcement and the H2D of id batches (what train_on_ids adds per step)
    t0 = time.perf_counter()
    for k in range(n):
        u, h, c = dev[k % len(dev)]
        _, h2, c2 = dev[(k + 1) % len(dev)]
        model.prefetch_ids(h2, c2)
        model._train_step(u, h, c, True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('with prefetch_ids: host enqueue %.1f us/step; wall incl. drain %.1f us/step' % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    staged = None
    t0 = time.perf_counter()
    for k in range(n):
        staged = model.train_on_ids(*batches[k % len(batches)], next_ids=batches[(k + 1) % len(batches)], staged=staged)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('train_on_ids (numpy ids -> H2D -> step): host %.1f us/step; wall incl. drain %.1f us/step' % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    t0 = time.perf_counter()
    for k in range(n):
        x = model._ids_to_dev(batches[k % len(batches)])
    t1 = time.perf_counter()
    print('_ids_to_dev alone: %.1f us/call' % ((t1 - t0) / n * 1e6))
