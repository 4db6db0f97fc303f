"""Two (and four) REAL ranks (one process each, torch.distributed over `gloo` -- RCCL refuses two ranks on one device) drive the
data-parallel training step of NativeModel on the one GPU of the test box: rank-sharded batches, the ONE all-reduce whose
tail carries the sparse user rows (both tables for iigru), Adam over the exchanged rows.  Checks: the replicas stay
bit-identical, and (dropout off) the parameters after three steps equal a single process training on the concatenated
batches, up to the summation order of the two half-batch means.  The reference is single-device; DESIGN.md section 5."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import torch.distributed as dist
from conftest import make_batch
from test_gpu_model import MID, build
arch, out, port, nparts = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', port
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
B, W, T, K1 = 8, 6, 9, 5
O, p, model, rng = build(arch, MID, B if world > 1 else nparts * B, W, T, K1, 0.0)
r2 = np.random.default_rng(42)
for step in range(3):
    parts = [make_batch(r2, B, W, T, K1, MID['V'], MID['N_u']) for _ in range(nparts)]
    parts[1][0][0] = parts[0][0][1]                    # one user on two ranks
    if world > 1:
        user, hist, cands = parts[rank]
    else:
        user, hist, cands = (np.concatenate(x) for x in zip(*parts))
    X = [user.astype(np.int64), hist.astype(np.float64)] + [cands[:, j].astype(np.float64) for j in range(K1)]
    model.train_on_batch(X, None)
torch.cuda.synchronize()
w = model.get_weights()
if world > 1:
    from mnexp_amd import dist as D
    dev = D.replica_checksum([torch.from_numpy(np.ascontiguousarray(a)).double() for a in w.values()])
    assert float(dev.max()) == 0.0, dev
if rank == 0:
    np.savez(out, **w)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


@pytest.mark.parametrize('arch,ranks,lazy', [('igru', 2, None), ('iigru', 2, None), ('igru', 4, None), ('igru', 2, '1'),
                                             ('iigru', 2, '1')])
def test_two_process_data_parallel_equals_one_process_on_the_joint_batch(arch, ranks, lazy, tmp_path):
    """lazy = '1': the lazily updated user table (large tables; forced here) -- rows another rank touched take their
    pending zero-gradient updates inside the row step of the exchange."""
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    code = _CHILD % (ROOT, os.path.join(ROOT, 'tests'))
    port = str(29600 + (os.getpid() % 300))
    outs = {}
    for world in (ranks, 1):
        out = str(tmp_path / ('w%d.npz' % world))
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
            if lazy:
                env['LSTUR_LAZY_ADAM'] = lazy
            procs.append(subprocess.Popen([sys.executable, '-c', code, arch, out, port, str(ranks)], env=env, cwd=ROOT,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for pr in procs:
            text = pr.communicate(timeout=600)[0].decode()
            assert pr.returncode == 0, text[-3000:]
        outs[world] = np.load(out)
    for n in outs[1].files:
        a, b = outs[ranks][n].astype(np.float64), outs[1][n].astype(np.float64)
        scale = max(np.abs(b).max(), 1e-6)
        assert np.abs(a - b).max() <= 2e-5 * scale + 1e-7, (n, np.abs(a - b).max(), scale)
