"""LSTUR training throughput on MI355X: training impressions/sec (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = forward + backward + (N > 1: ONE RCCL all-reduce of the dense-gradient bucket whose tail carries every
rank's sparse user-row gradients, mnexp_amd/dist.py) + Keras-Adam over the dense bucket and the whole user table, on one
batch of synthetic doc-id samples that is already resident in HBM.  Default workload = BASELINE.json
configs[1]: LSTUR-ini (arch igru), 300-d word embeddings, title_len 30, hist_len 50, 1+4 candidates,
batch 64 per GPU (weak scaling), dropout 0.2 (reference default), 50k users / 20k docs / 50k vocab.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1] / configs[2]
    'lstur_ini_b64': dict(arch='igru', B=64, W=50, T=30, K1=5, D=300, F=400, k=3, U=200, V=50000, N_u=50000, N_d=20000),
    'lstur_con_b64': dict(arch='gru', B=64, W=50, T=30, K1=5, D=300, F=400, k=3, U=200, V=50000, N_u=50000, N_d=20000),
    # BASELINE.json configs[3]
    'lstur_ini_b512_1Musers': dict(arch='igru', B=512, W=50, T=30, K1=5, D=300, F=400, k=3, U=200, V=100000,
                                   N_u=1000000, N_d=200000),
    # BASELINE.json configs[4]: the build-defined multi-head self-attention news encoder (16 heads x 16; the reference has
    # no such encoder) at the config-2 and config-4 shapes
    'mhsa_b64': dict(arch='igru', B=64, W=50, T=30, K1=5, D=300, F=256, k=1, U=200, V=50000, N_u=50000, N_d=20000,
                     encoder='mhsa', heads=16),
    'mhsa_b512_1Musers': dict(arch='igru', B=512, W=50, T=30, K1=5, D=300, F=256, k=1, U=200, V=100000, N_u=1000000,
                              N_d=200000, encoder='mhsa', heads=16),
}
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured copy ceiling)


def synth_inputs(wl, rng, n_batches):
    """Synthetic resident inputs: title table [N_d, T] (ragged, zero padded, doc 0 = pad), word table,
    and n_batches of (user, hist_ids, cand_ids) with ~10 % front-padded histories."""
    T, N_d, V = wl['T'], wl['N_d'], wl['V']
    titles = rng.integers(1, V, size=(N_d, T), dtype=np.int32)
    lens = rng.integers(5, T + 6, size=N_d)
    titles[np.arange(T)[None, :] >= lens[:, None]] = 0
    titles[0] = 0
    word = rng.normal(0.0, 0.1, size=(V, wl['D'])).astype(np.float32)
    word[0] = 0.0
    batches = []
    for _ in range(n_batches):
        user = rng.integers(0, wl['N_u'], size=wl['B'], dtype=np.int32)
        hist = rng.integers(1, N_d, size=(wl['B'], wl['W']), dtype=np.int32)
        npad = np.where(rng.random(wl['B']) < 0.1, rng.integers(1, wl['W'], size=wl['B']), 0)
        hist[np.arange(wl['W'])[None, :] < npad[:, None]] = 0
        cands = rng.integers(1, N_d, size=(wl['B'], wl['K1']), dtype=np.int32)
        batches.append((user, hist, cands))
    return titles, word, batches


def rows_covered(mask):
    """padded embedding rows tp = 0 .. T + 1 of a title that the 3-row window of some non-pad token covers (what the
    split-layout gather writes and the conv kernels read); mask [..., T] = token is not a pad"""
    T_ = mask.shape[-1]
    q = np.pad(mask, [(0, 0)] * (mask.ndim - 1) + [(2, 2)])
    return int((q[..., 0:T_ + 2] | q[..., 1:T_ + 3] | q[..., 2:T_ + 4]).sum())


def pmc_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/;
    FETCH_SIZE doubled per the gfx950 correction) -- None when no file matches the workload."""
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, 'profiles'))) if os.path.isdir(os.path.join(ROOT, 'profiles')) else []:
        if name.endswith('pmc_traffic.json'):
            try:
                d = json.load(open(os.path.join(ROOT, 'profiles', name)))
            except Exception:
                continue
            if d.get('workload') == workload:
                best = d
    return best


def kernel_sources_hash():
    """Hash of the kernel sources: the PMC traffic file records it; a mismatch means the passes predate the kernels."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, 'mnexp_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()[:16]


def cpu_baseline(wl, titles, word, batches, budget_s=15.0):
    """The NumPy oracle (kind 'port'; the reference's Keras/TF stack cannot run here) timed on the host
    cores for a bounded number of full training steps of the SAME workload in fp32."""
    from oracle import lstur_oracle as O
    rng = np.random.default_rng(0)
    enc = wl.get('encoder', 'cnnatt')
    p = O.init_params(rng, V=wl['V'], D=wl['D'], F=wl['F'], k=wl['k'], U=wl['U'], N_u=wl['N_u'], arch=wl['arch'],
                      word_emb=word, dtype=np.float32, encoder=enc, heads=wl.get('heads', 1))
    opt = O.Adam(p, O.dense_param_names(wl['arch'], encoder=enc) + ['user_emb'], lr=1e-3)
    threads, limiter = usable_cpus(), None
    try:
        import threadpoolctl
        blas = max(i['num_threads'] for i in threadpoolctl.threadpool_info() if i.get('user_api') == 'blas')
        threads = min(threads, blas)
        limiter = threadpoolctl.threadpool_limits(limits=threads, user_api='blas')
    except Exception:
        pass
    steps, t0 = 0, time.perf_counter()
    while True:
        user, hist, cands = batches[steps % len(batches)]
        O.train_step(p, wl['arch'], opt, user, titles[hist], titles[cands], drop_p=0.2, seed=0, step=steps)
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 8:
            break
    if limiter is not None:
        limiter.restore_original_limits()
    out = dict(value=steps * wl['B'] / el, unit='impressions/s', cores=int(threads), kind='port',
               sample='%d full training steps (fwd+bwd+Adam, fp32 NumPy/BLAS) of the same workload, %.1f s'
                      % (steps, el), host_cpus=os.cpu_count(), usable_cpus=usable_cpus(), cpu_model=cpu_model_string(), numpy_value=steps * wl['B'] / el)
    if wl.get('encoder', 'cnnatt') == 'cnnatt':
        # the torch-CPU fp32 step on all cores beside the NumPy port; `value` is the FASTER of the two
        tc = cpu_baseline_torch(wl, titles, word, batches)
        out['torch_cpu'] = tc
        if tc['value'] > out['value']:
            out.update(value=tc['value'], cores=tc['threads'], sample=tc['sample'])
    return out


def usable_cpus():
    """Cores this process may actually run on: the affinity mask and the cgroup CPU quota (a GPU box hands one GPU's job
    a share of the host, e.g. 16 of 256 CPUs -- sizing thread pools by os.cpu_count() there oversubscribes 16x)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_model_string():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline_torch(wl, titles, word, batches, budget_s=12.0):
    """The same training step in torch-CPU fp32 on ALL host cores: F.conv1d, autograd through the masked GRU
    loop (oracle/torch_ref.py, the torch restatement the oracle is cross-checked against), dense Keras-Adam over every
    parameter incl. the whole user table.  Dropout masks from torch's generator (throughput only)."""
    import torch
    from oracle import lstur_oracle as O, torch_ref as TR
    nthreads = usable_cpus()
    torch.set_num_threads(nthreads)
    enc = wl.get('encoder', 'cnnatt')
    p = O.init_params(np.random.default_rng(0), V=wl['V'], D=wl['D'], F=wl['F'], k=wl['k'], U=wl['U'], N_u=wl['N_u'],
                      arch=wl['arch'], word_emb=word, dtype=np.float32, encoder=enc, heads=wl.get('heads', 1))
    names = O.dense_param_names(wl['arch'], encoder=enc) + ['user_emb']
    tp = TR.to_t(p, dtype=torch.float32, grad_names=names)
    m = {n: torch.zeros_like(tp[n]) for n in names}
    v = {n: torch.zeros_like(tp[n]) for n in names}
    b1, b2, eps, lr, drop = 0.9, 0.999, 1e-7, 1e-3, 0.2
    B, W, T, K1 = wl['B'], wl['W'], wl['T'], wl['K1']
    nt = B * (W + K1)
    steps, t0 = 0, time.perf_counter()
    while True:
        user, hist, cands = batches[steps % len(batches)]
        keep1 = (torch.rand(nt, T, wl['D']) >= drop)
        keep2 = (torch.rand(nt, T, wl['F']) >= drop)
        loss, _, _ = TR.model_loss(tp, wl['arch'], user, titles[hist], titles[cands], keep1, keep2, 1.0 / (1.0 - drop))
        loss.backward()
        t = steps + 1
        lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        with torch.no_grad():
            for n in names:          # (the embedding gradient arrives dense: the sweep covers the whole table, like TF-1 Keras)
                g = tp[n].grad
                m[n].mul_(b1).add_(g, alpha=1 - b1)
                v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
                tp[n].sub_(lr_t * m[n] / (v[n].sqrt() + eps))
                tp[n].grad = None
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= 8:
            break
    return dict(value=steps * B / el, threads=int(torch.get_num_threads()),
                sample='%d full training steps (fwd+autograd bwd+dense Adam, fp32 torch-CPU %s) of the same workload, %.1f s'
                       % (steps, torch.__version__, el))


def e2e_legs(wl, word, steps, dropout, device):
    """End-to-end throughput of the drop-in path: files in the reference's formats (DocMeta.tsv + ClickData.tsv of the
    workload's FULL shape -- all its users and docs -- written by the C++ generator of include/lstur_data.h) -> task loading
    with the C++ loaders (`task_load_s`: DocMeta + ClickData parsing, doc-id check) -> samples from (a) the C++ feed
    (--native-feed) and (b) the task's own Python generators in id form (--ids) -> host batch -> H2D -> step, ALL inside the
    timed region.  The samples are real click histories (mostly shorter than the window, so fewer non-pad tokens per step
    than the device-resident headline, whose histories are 90 % full)."""
    import contextlib
    import io
    import random
    import shutil
    import tempfile
    import torch
    from mnexp_amd import settings, task as T
    from mnexp_amd.native_model import NativeModel
    from mnexp_amd.data_native import synth_write
    d = tempfile.mkdtemp(prefix='lstur_bench_e2e_')
    try:
        t0 = time.perf_counter()
        synth_write(d, wl['N_d'], wl['V'], wl['T'], wl['N_u'], 20, 4, seed=0)
        write_s = time.perf_counter() - t0
        open(os.path.join(d, 'Vocab.tsv'), 'w').close()          # (the word table comes from the bench: same shape)
        settings.public_config = dict(input_training_data_path=d, input_validation_data_path=d,
                                      input_previous_model_path=d, output_model_path=d, log_dir=d)
        base = dict(task='Seq2VecPaperSoftmaxId', arch=wl['arch'], batch_size=wl['B'], window_size=wl['W'],
                    title_shape=wl['T'], body_shape=4, negative_samples=wl['K1'] - 1, training_step=steps,
                    validation_step=1, max_impression=200, news_encoder='cnnatt', textual_embedding_dim=wl['D'], days=30,
                    debug=False, epochs=1, validation_impression=10, testing_impression=10, learning_rate=1e-3,
                    learning_rate_decay=0.5, dropout=dropout, user_embedding_dim=wl['U'],
                    title_filter_shape=(wl['F'], wl['k']), score_model='dot', enable_pretrain_encoder=False,
                    textual_embedding_trainable=False, name='', gain=1.0, seed=1, ids=True, native_feed=True,
                    native_load=True, background=True, pretrain_name='')
        with contextlib.redirect_stdout(io.StringIO()):
            cfg = settings.Config(base)
        t0 = time.perf_counter()
        h = T.get(cfg)
        load_s = time.perf_counter() - t0
        model = NativeModel(word, len(h.data), wl['B'], wl['W'], wl['T'], wl['K1'] - 1, wl['F'], wl['k'], wl['U'],
                            arch=wl['arch'], dropout=dropout, lr=1e-3, seed=7, device=device,
                            title_table=h.docs.title_table)
        out = dict(users=len(h.data), docs=wl['N_d'], task_load_s=round(load_s, 2), synth_write_s=round(write_s, 1),
                   clickdata_mb=round(os.path.getsize(os.path.join(d, 'ClickData.tsv')) / 1e6, 1), steps=steps,
                   loader='C++ (liblstur_data.so: lstur_docs_load + lstur_clicks_load incl. the doc-id check)')
        np.random.seed(0)
        random.seed(0)
        for name, feed in (('native_feed', h.native_feed()), ('python_ids_feed', h.train_ids)):
            nonpad = []
            # one batch of look-ahead, as NativeModel.fit_generator does: the next batch is uploaded and announced to the
            # library (lstur_prefetch_ids) before the current step is launched
            # ... and, like Keras' fit_generator (workers=1, max_queue_size=10) and NativeModel.fit_generator, the generator
            # runs in one background thread at most 10 batches ahead
            from mnexp_amd.native_model import prefetched
            feed = prefetched(feed, steps + 12, 10)
            X, _ = next(feed)
            staged = None
            for _ in range(10):
                Xn, _ = next(feed)
                staged = model.train_on_ids(*X, next_ids=Xn, staged=staged)
                X = Xn
            torch.cuda.synchronize()
            seen = []
            t0 = time.perf_counter()
            for _ in range(steps):
                Xn, _ = next(feed)
                staged = model.train_on_ids(*X, next_ids=Xn, staged=staged)
                seen.append(X)
                X = Xn
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            next(feed, None)                     # (lets the worker finish its count)
            nonpad = [int((h.docs.title_table[x[1]] != 0).sum() + (h.docs.title_table[x[2]] != 0).sum()) for x in seen]
            out[name] = dict(value=round(wl['B'] * steps / el, 1), unit='impressions/s wall (feed + H2D + step)',
                             ms_per_step=round(1e3 * el / steps, 4),
                             non_pad_token_fraction=round(float(np.mean(nonpad)) / (wl['B'] * (wl['W'] + wl['K1']) * wl['T']), 4))
        del model
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def parity_check(model, wl, titles, batch):
    """Outside the timed region: logits of one bench batch (dropout off) from the HIP path against the fp64 CPU oracle with
    the model's CURRENT (trained) weights -- the number that shows the split-operand conv kernels are fp32-grade at the bench
    shapes (BASELINE.json: fp32 logits within 1e-3 relative)."""
    import torch
    from oracle import lstur_oracle as O
    from mnexp_amd._lib import lib, check
    user, hist, cands = batch
    tt = titles
    u, h, c = model._dev_i32(user), model._dev_i32(tt[hist]), model._dev_i32(tt[cands])
    check(lib.lstur_evaluate(model._ctx, u.data_ptr(), h.data_ptr(), c.data_ptr(), int(user.shape[0]),
                             model.stats.data_ptr(), model._stream()), 'lstur_evaluate')
    torch.cuda.synchronize()
    got = model.peek('logits').cpu().numpy().reshape(len(user), -1).astype(np.float64)
    p = {k: np.asarray(v, dtype=np.float64) for k, v in model.get_weights().items()}
    p['word_emb'] = model.word_emb.cpu().numpy().astype(np.float64)
    if wl.get('encoder') == 'mhsa':
        p['mhsa_heads'] = np.int64(wl['heads'])
    ref = O.forward_backward(p, wl['arch'], user, tt[hist], tt[cands], drop_p=0.0, want_grads=False)['logits']
    return dict(max_rel_logit_err_vs_fp64_oracle=float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)),
                logit_scale=float(np.abs(ref).max()), samples=int(len(user)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--e2e-steps', type=int, default=150,
                    help='steps of the end-to-end legs (sample feed + H2D inside the timed region; 0: skip)')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='lstur_ini_b64', choices=sorted(WORKLOADS))
    ap.add_argument('--dropout', type=float, default=0.2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prefetch', action='store_true', help='stage every batch inside its own step (no lstur_prefetch_ids)')
    ap.add_argument('--no-profile', action='store_true', help='skip the HIP-event brackets around kernels')
    ap.add_argument('--fresh-users', action='store_true',
                    help='draw every step\'s user ids uniformly from the whole table (default: the 32 synthetic batches cycle, '
                         'i.e. 32 x B distinct users) -- the steady state of the lazy table update at 1 M users needs this')
    ap.add_argument('--force-dp', action='store_true', help='diagnostic: run the RCCL collectives of the data-parallel step even with one rank')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mnexp_amd.native_model import NativeModel

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)' % (args.gpus, args.gpus, world))
    # LSTUR_BENCH_BACKEND=gloo: rehearsal of the N > 1 launch on a box with fewer GPUs than ranks (RCCL refuses two ranks on
    # one device; gloo moves the bucket through the host): same code path, meaningless throughput
    backend = os.environ.get('LSTUR_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1 or args.force_dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = WORKLOADS[args.workload]
    rng = np.random.default_rng(1234 + rank)
    titles, word, _ = synth_inputs(wl, np.random.default_rng(1234), 0)   # identical tables on every rank
    _, _, batches = synth_inputs(wl, rng, 32)                            # per-rank sample stream
    model = NativeModel(word, wl['N_u'], wl['B'], wl['W'], wl['T'], wl['K1'] - 1, wl['F'], wl['k'], wl['U'],
                        arch=wl['arch'], dropout=args.dropout, lr=1e-3, seed=7, device='cuda:%d' % local,
                        title_table=titles, news_encoder=wl.get('encoder', 'cnnatt'), heads=wl.get('heads', 0))
    model.force_collectives = args.force_dp
    dev = [(model._dev_i32(u), model._dev_i32(h), model._dev_i32(c)) for u, h, c in batches]
    torch.cuda.synchronize()

    gen = torch.Generator(device='cuda:%d' % local).manual_seed(99 + rank)

    def users_of(k):
        if args.fresh_users:
            return torch.randint(0, wl['N_u'], (wl['B'],), dtype=torch.int32, device='cuda:%d' % local, generator=gen)
        return dev[k % len(dev)][0]

    def run(nsteps, offset):
        u = users_of(offset)
        for i in range(nsteps):
            _, h, c = dev[(offset + i) % len(dev)]
            u2 = users_of(offset + i + 1)
            if not args.no_prefetch and i + 1 < nsteps:      # announce the next batch: its staging overlaps this step
                _, h2, c2 = dev[(offset + i + 1) % len(dev)]
                model.prefetch_ids(h2, c2, u2)
            model._train_step(u, h, c, True)
            u = u2

    run(args.warmup, 0)
    torch.cuda.synchronize()
    if not args.no_profile:
        model.profile(2)          # timed region: HIP-event brackets around the roofline kernels only (each bracket costs ~6 us)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    loss, acc = (model.stats_acc / max(1, model.step)).tolist()

    prof = {} if args.no_profile else model.profile_read()
    model.profile(False)
    prof_steps = args.steps
    if not args.no_profile:       # the full per-family breakdown from a separate UNTIMED run of steps
        nb = min(20, args.steps)
        model.profile(1)
        run(nb, args.warmup + args.steps)
        torch.cuda.synchronize()
        full = model.profile_read()
        model.profile(False)
        for k, (ms, n) in full.items():
            if k not in prof:
                prof[k] = (ms * prof_steps / nb, n)
    if rank == 0:
        n_tok = wl['B'] * (wl['W'] + wl['K1']) * wl['T']
        # the conv kernels run over the non-pad tokens only (pad outputs are zero by definition); `achieved`
        # counts the flops actually executed, `effective_algorithmic` the reference's full per-token count
        real_tok = float(np.mean([(titles[h] != 0).sum() + (titles[c] != 0).sum() for _, h, c in batches]))
        conv_flop_full = 2.0 * n_tok * wl['k'] * wl['D'] * wl['F']        # 720 kFLOP/token at the defaults
        conv_flop = conv_flop_full * real_tok / n_tok
        gather_bytes = n_tok * (4 + 2 * 4 * wl['D'])                       # 2404 B/token stand-alone
        # per STEP (a family may launch more than once per step: the two-lane step, the conv tail fix-up)
        kern = {k: dict(avg_us=1e3 * ms / max(1, args.steps), launches=n) for k, (ms, n) in prof.items()}
        roofline = None
        roofline_gather = None
        pmc = pmc_traffic(args.workload)
        traffic = (lambda k: pmc['kernels'][k]['traffic_bytes'] if pmc and k in pmc['kernels'] else None)
        # the PMC passes are a committed file: stale when the kernel sources changed since (tools/pmc_traffic.py stamps it)
        traffic_stale = bool(pmc) and pmc.get('kernel_sources_hash') != kernel_sources_hash()
        mhsa = wl.get('encoder') == 'mhsa'
        if prof and mhsa:
            # token stage = Q|K|V projection GEMMs over ALL tokens (pads included) + the attention kernels
            dom = max(('qkv_proj', 'qkv_wgrad'), key=lambda k: kern[k]['avg_us'])
            flop = 2.0 * n_tok * wl['D'] * 3 * wl['F']
            peak = PEAK_BF16_MFMA_TFLOPS if model.mhsa_proj_dtype == 'bf16' else PEAK_F32_MFMA_TFLOPS
            ach = flop / (kern[dom]['avg_us'] * 1e-6) / 1e12
            roofline = dict(kernel=dom, bound='mfma', achieved=round(ach, 2), peak=peak, unit='TFLOP/s',
                            frac=round(ach / peak, 4), traffic=traffic(dom), avg_us=round(kern[dom]['avg_us'], 1),
                            flop_per_launch=flop, operand_dtype=model.mhsa_proj_dtype)
        if prof and not mhsa:
            dom = max(('conv_fwd', 'conv_wgrad'), key=lambda k: kern[k]['avg_us'])
            dom_flop = conv_flop
            side = None
            if dom == 'conv_wgrad' or 'conv_wgrad_side' in kern:
                # the kernel gradient is two launches when the candidate titles' share runs beside the recurrence
                # (family conv_wgrad_side, off the critical path): each is priced on the tokens it contracts over
                hist_tok = float(np.mean([(titles[h] != 0).sum() for _, h, c in batches]))
                if 'conv_wgrad_side' in kern:
                    side_flop = conv_flop * (real_tok - hist_tok) / real_tok
                    side = dict(kernel='conv_wgrad_side', avg_us=round(kern['conv_wgrad_side']['avg_us'], 1), flop_per_launch=side_flop,
                                note='candidate titles\' tokens: a launch of its own on the side stream under the recurrence\'s backward')
                    if dom == 'conv_wgrad':
                        dom_flop = conv_flop * hist_tok / real_tok
            ach = dom_flop / (kern[dom]['avg_us'] * 1e-6) / 1e12
            # split-operand kernels (convx.hip): every fp32-grade product is THREE fp16 (f16x2) or SIX bf16 (bf16x3) MFMA
            # products, so the kernel's roofline is the dense 16-bit matrix peak / 3 or / 6; the exact-fp32 MFMA kernels
            # (LSTUR_CONV_F32=1) are priced against the fp32 peak
            parts = {'f16x2': 2, 'bf16x3': 3}.get(model.conv_scheme)
            products = {'f16x2': 3.0, 'bf16x3': 6.0}.get(model.conv_scheme, 1.0)
            peak = PEAK_BF16_MFMA_TFLOPS / products if parts else PEAK_F32_MFMA_TFLOPS
            scheme = {'f16x2': 'f16x2: both fp32 operands as two fp16 parts of the power-of-two scaled value, 3 MFMA products, '
                               'fp32 accumulate (fp32-grade; peak = 2500 TFLOP/s dense fp16 / 3)',
                      'bf16x3': 'bf16x3: 3-way bf16 split of both fp32 operands, 6 MFMA products, fp32 accumulate '
                                '(fp32-grade; peak = 2500 TFLOP/s dense bf16 / 6)'}.get(model.conv_scheme, 'fp32 MFMA 16x16x4')
            roofline = dict(kernel=dom, bound='mfma', achieved=round(ach, 2), peak=round(peak, 1), unit='TFLOP/s',
                            frac=round(ach / peak, 4), traffic=traffic(dom),
                            avg_us=round(kern[dom]['avg_us'], 1), flop_per_launch=dom_flop,
                            non_pad_token_fraction=round(real_tok / n_tok, 4),
                            effective_algorithmic=round(conv_flop_full * dom_flop / conv_flop / (kern[dom]['avg_us'] * 1e-6) / 1e12, 2),
                            scheme=scheme, mfma_executed_tflops=round(ach * products, 1),
                            frac_vs_fp32_mfma_peak=round(ach / PEAK_F32_MFMA_TFLOPS, 4))
            if side:
                side['frac'] = round(side['flop_per_launch'] / (side['avg_us'] * 1e-6) / 1e12 / peak, 4)
                roofline['side_launch'] = side
            if parts:
                # what the kernel has to move at least: every embedding row some non-pad token's window covers once (its
                # 16-bit parts), the split weight once, and its own output / input gradient rows (fp32) once
                e_rows = sum(rows_covered(titles[x] != 0) for b_ in batches for x in (b_[1], b_[2])) / float(len(batches))
                dp = (wl['D'] + 7) // 8 * 8
                alg = e_rows * dp * 2 * parts + parts * wl['k'] * dp * ((wl['F'] + 31) // 32 * 32) * 2 + real_tok * wl['F'] * 4
                roofline['traffic_algorithmic'] = int(alg * (dom_flop / conv_flop))
                if roofline['traffic']:
                    roofline['traffic_over_algorithmic'] = round(roofline['traffic'] / roofline['traffic_algorithmic'], 2)
        if prof:
            # The gather is priced on the bytes the SELECTED kernel moves (ids + table rows read + rows written in its layout):
            # `achieved` / `frac` can therefore not exceed the HBM peak.  The SURVEY's stand-alone figure (2404 B/token: id +
            # fp32 row in + fp32 row out for EVERY token, pads included) over the same launch time is kept as `effective_*`:
            # rows the kernel skips, or writes in fewer bytes, count there as moved.
            us = kern['gather']['avg_us']
            n_titles = wl['B'] * (wl['W'] + wl['K1'])
            if model.conv_scheme in ('f16x2', 'bf16x3'):
                parts = {'f16x2': 2, 'bf16x3': 3}[model.conv_scheme]
                rows = sum(rows_covered(titles[x] != 0) for b in batches for x in (b[1], b[2])) / float(len(batches))
                moved = rows * (4 * wl['D'] + 2 * parts * ((wl['D'] + 7) // 8 * 8)) + n_tok * 4
                layout = 'Ex: %d x 16-bit parts per element (%d B), padded rows no window of a non-pad token covers skipped' % (parts, 2 * parts)
            elif mhsa:
                moved = n_tok * (4 + 4 * wl['D'] + 2 * wl['D'])
                layout = 'bf16 rows [N_tok, D] for the projections: fp32 row in, bf16 row out, every token'
            else:
                moved = n_titles * (wl['T'] + wl['k'] - 1) * 4 * wl['D'] + n_tok * (4 + 4 * wl['D'])
                layout = 'fp32 padded rows [N_t, T + k - 1, D]: every row written, every token\'s table row read'
            gb = moved / (us * 1e-6) / 1e9
            eff = gather_bytes / (us * 1e-6) / 1e9
            roofline_gather = dict(kernel='gather', bound='hbm', achieved=round(gb, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                                   frac=round(gb / PEAK_HBM_GBS, 4), traffic=traffic('gather'), avg_us=round(us, 1),
                                   bytes_per_launch=int(moved), layout=layout,
                                   effective_bytes_per_launch=gather_bytes, effective_gbs=round(eff, 1),
                                   effective_frac=round(eff / PEAK_HBM_GBS, 4),
                                   effective_note='SURVEY 8d: 2404 B/token stand-alone (id + fp32 row read + fp32 row write, pads '
                                                  'included) over the same launch time')
        out = {
            'metric': 'training impressions/sec', 'value': round(world * wl['B'] * args.steps / el, 1),
            'unit': 'impressions/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * el / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if wl.get('encoder') == 'mhsa' else 'f32', 'data': 'synthetic',
            'conv_scheme': model.conv_scheme,
            'config': {'workload': args.workload, 'arch': wl['arch'], 'batch_per_gpu': wl['B'],
                       'global_batch': wl['B'] * world, 'hist_len': wl['W'], 'title_len': wl['T'],
                       'candidates': wl['K1'], 'word_dim': wl['D'], 'filters': wl['F'], 'user_dim': wl['U'],
                       'n_users': wl['N_u'], 'n_docs': wl['N_d'], 'vocab': wl['V'], 'dropout': args.dropout,
                       'news_encoder': wl.get('encoder', 'cnnatt'), 'heads': wl.get('heads', 0),
                       'parallelism': 'dp%d' % world, 'collective_backend': ('rccl' if backend == 'nccl' else backend) if world > 1 else None,
                       'optimizer': 'keras-adam (dense user-table update, applied lazily per row: bit-identical)' if model.lazy_table
                                    else 'keras-adam (dense user-table sweep)',
                       'lazy_table': bool(model.lazy_table), 'fresh_users': bool(args.fresh_users),
                       'prefetch': not args.no_prefetch,
                       # the synthetic histories are 90 % full windows (62 % non-pad tokens): heavier than real click
                       # histories (the `e2e` legs run at ~49 % non-pad tokens)
                       'history_fill': 'synthetic: 90 %% of the samples have a full %d-slot history' % wl['W']},
            'roofline': roofline, 'roofline_gather': roofline_gather, 'traffic_stale': traffic_stale,
            'kernels_us': {k: round(v['avg_us'], 1) for k, v in sorted(kern.items())},
            'train_loss': round(loss, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            out['parity'] = parity_check(model, wl, titles, batches[0])
            if args.e2e_steps > 0 and wl.get('encoder', 'cnnatt') == 'cnnatt':
                out['e2e'] = e2e_legs(wl, word, args.e2e_steps, args.dropout, 'cuda:%d' % local)
            out['cpu_baseline'] = cpu_baseline(wl, titles, word, batches)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1 or args.force_dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
