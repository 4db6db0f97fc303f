"""Data-parallel host logic on CPU: world_size-2 gloo, with the oracle standing in for the GPU compute.
Checks that (1) averaging per-rank gradients of half batches through dist.allreduce_bucket equals the
global-batch gradient, (2) all-gathered sparse user rows reproduce the global densified gradient with
duplicates across ranks, (3) replicas stay bit-identical after the update, (4) users are partitioned."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

DIMS = dict(V=60, D=12, F=16, k=3, U=8, N_u=11)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from conftest import make_batch
    from mnexp_amd import dist as D
    from oracle import lstur_oracle as O
    rng = np.random.default_rng(3)
    p = O.init_params(rng, arch='igru', dtype=np.float64, **DIMS)
    B, W, T, K1 = 8, 4, 7, 3
    user, hist, cands = make_batch(rng, B, W, T, K1, DIMS['V'], DIMS['N_u'])
    user[B // 2] = user[0]                       # the same user on two ranks
    names = O.dense_param_names('igru')
    # --- single-process reference on the global batch
    ref = O.forward_backward(p, 'igru', user, hist, cands)
    ref_user = O.densify_user_grad(*p['user_emb'].shape, ref['user_rows'], ref['user_row_grads'], np.float64)
    # --- this rank's half, gradient of the LOCAL mean pre-scaled by 1/world (what the library does)
    sl = slice(rank * B // world, (rank + 1) * B // world)
    loc = O.forward_backward(p, 'igru', user[sl], hist[sl], cands[sl])
    flat = torch.from_numpy(np.concatenate([loc['grads'][n].ravel() for n in names]) / world)
    D.allreduce_bucket(flat, prescaled=True)
    got = flat.numpy()
    want = np.concatenate([ref['grads'][n].ravel() for n in names])
    assert np.abs(got - want).max() < 1e-12
    rows = torch.from_numpy(loc['user_rows'].astype(np.int32))
    rg = torch.from_numpy(loc['user_row_grads'] / world)
    all_rows, all_grads = D.allgather_rows(rows, rg)
    assert all_rows.shape[0] == B and np.array_equal(all_rows.numpy(), user)
    dense = O.densify_user_grad(*p['user_emb'].shape, all_rows.numpy(), all_grads.numpy(), np.float64)
    assert np.abs(dense - ref_user).max() < 1e-12
    # --- identical update on every rank
    g = {n: a for n, a in zip(names, np.split(got, np.cumsum([p[n].size for n in names])[:-1]))}
    g = {n: g[n].reshape(p[n].shape) for n in names}
    g['user_emb'] = dense
    opt = O.Adam(p, names + ['user_emb'])
    opt.step(p, g)
    dev = D.replica_checksum([torch.from_numpy(p[n]) for n in names + ['user_emb']])
    assert float(dev.max()) == 0.0
    owned = list(D.shard_users(DIMS['N_u'], rank, world))
    t = torch.zeros(DIMS['N_u'], dtype=torch.int64)
    t[owned] = 1
    dist.all_reduce(t)
    assert bool((t == 1).all())                  # a partition: every user on exactly one rank
    out.put((rank, 'ok'))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_n_rank_gradient_equivalence(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(world)) == list(range(world))


def test_task_shards_users_across_ranks(monkeypatch):
    from test_data_path import make_config, GOLD
    from mnexp_amd import dist as D, task
    cfg = make_config(os.path.join(GOLD, 'tiny'))
    h = task.get(cfg)
    seen = []
    for r in range(3):
        monkeypatch.setattr(D, 'world_size', lambda group=None: 3)
        monkeypatch.setattr(D, 'rank', lambda group=None, r=r: r)
        seen.append([u for u, _ in h._users(shard=True)])
        # validation / test generators walk EVERY user on every rank: the reported metrics are the single-GPU ones
        assert [u for u, _ in h._users()] == list(range(len(h.data)))
    assert sorted(sum(seen, [])) == list(range(len(h.data)))
    assert seen[1][:3] == [1, 4, 7]
    # the id-less task shards the same way (or every rank would train on the same samples), user column = None
    cfg2 = make_config(os.path.join(GOLD, 'tiny'), task='Seq2VecPaperSoftmax', arch='gru')
    h2 = task.get(cfg2)
    monkeypatch.setattr(D, 'rank', lambda group=None: 2)
    got = list(h2._users(shard=True))
    assert [d is h2.data[i] for (u, d), i in zip(got, range(2, len(h2.data), 3))] == [True] * len(got)
    assert all(u is None for u, _ in got)


class _HashModel:
    """stands in for the GPU model: a deterministic score per (user, history, candidate) row"""

    class _Opt:
        lr = 1e-3

    optimizer = _Opt()

    def predict_ids(self, user, hist, cand):
        h = (user.astype(np.int64) * 7919 + hist.astype(np.int64).sum(axis=1) * 104729 + cand.astype(np.int64).reshape(-1) * 1299709) % 1000003
        return (h / 1000003.0).astype(np.float32)


def _eval_worker(rank, world, port, native, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from test_data_path import make_config, GOLD
    from mnexp_amd import task
    h = task.get(make_config(os.path.join(GOLD, 'tiny'), native_load=native, native_feed=native, validation_impression=37))
    h.model = h.test_model = _HashModel()
    seen = [i for i, _, _ in h.test_items(37, rank, world)]
    assert all(i % world == rank for i in seen) and len(seen) in (37 // world, 37 // world + 1)
    h.callback(0)
    out.put((rank, h.last_metrics, len(seen)))
    dist.destroy_process_group()


@pytest.mark.parametrize('native', [False, True])
def test_evaluation_is_sharded_by_impression_and_reduces_to_the_single_process_metrics(native):
    """SURVEY.md 8e / reference callback task/paper.py:497-524: under data parallelism rank r predicts validation impressions
    r, r + N, ... only; the all-reduced (sum of metric, count) gives every rank the single-process AUC / nDCG / MRR."""
    from test_data_path import make_config, GOLD
    from mnexp_amd import task
    h = task.get(make_config(os.path.join(GOLD, 'tiny'), native_load=native, native_feed=native, validation_impression=37))
    h.model = h.test_model = _HashModel()
    h.callback(0)
    want = h.last_metrics
    assert 0.0 < want['auc'] < 1.0
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, native, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = [q.get(timeout=5) for _ in range(world)]
    assert sum(n for _, _, n in got) == 37
    for _, m, _ in got:
        for k in want:
            assert abs(m[k] - want[k]) < 1e-12, (k, m[k], want[k])
