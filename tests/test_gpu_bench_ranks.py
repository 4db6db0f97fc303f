"""bench.py launched the way the driver launches it for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`): the ranks' barrier, the max-over-ranks clock, rank 0's ONE
JSON line.  The test box has one GPU and RCCL refuses two ranks on one device, so the ranks share cuda:0 and exchange over
gloo (LSTUR_BENCH_BACKEND=gloo) -- every line of the N > 1 path runs, the throughput means nothing."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('ranks', [2])
def test_bench_line_from_a_multi_rank_launch(ranks):
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    env = dict(os.environ, LSTUR_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--steps', '6', '--warmup', '2']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 alone prints
    out = json.loads(lines[0])
    assert out['n_gpus'] == ranks and out['steps'] == 6 and out['warmup'] == 2 and out['scaling'] == 'weak'
    assert out['config']['parallelism'] == 'dp%d' % ranks and out['config']['global_batch'] == ranks * out['config']['batch_per_gpu']
    assert out['value'] > 0 and abs(out['value'] - ranks * out['config']['batch_per_gpu'] * 6 / (out['ms_per_step'] * 6e-3)) < 1e-3 * out['value']
    assert out['roofline']['frac'] > 0 and out['cpu_baseline'] is None       # the CPU leg is rank 0 at N = 1 only
    assert out['train_loss'] == out['train_loss']                            # finite
