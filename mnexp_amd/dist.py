"""Data parallelism for the LSTUR path: one process per GPU, torch.distributed ('nccl' = RCCL over xGMI).

The reference is single-process; this is the build's own design:
  * users are partitioned by row number: rank r owns users r, r+N, r+2N, ... (user id = ClickData row,
    reference task/paper.py:531), every rank runs its own sample pool with the same local batch size;
  * ONE collective per step: an all-reduce of the flat gradient bucket (the library already scales by 1/N) whose
    tail carries the per-sample user-row gradients -- every rank writes its (row, grad) pairs into its own slot and
    zeroes the others, so the sum over ranks is the concatenation (lstur_dp_stage / lstur_dp_unpack, include/lstur.h);
    every rank then applies the identical dense Keras-Adam sweep to its full replica of the user table.
    (allgather_rows below is the general fallback: > 8 ranks or row numbers above 2^24.)
The helpers take plain tensors so the same code runs on 'gloo' (CPU tests) and 'nccl' (MI355X).
"""
import os

import torch
import torch.distributed as dist


def maybe_init_from_env():
    """Initialise the default group when launched by torch.distributed.run (WORLD_SIZE > 1)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 or dist.is_initialized():
        return world
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if torch.cuda.is_available():
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        dist.init_process_group('gloo')
    return world


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def shard_users(n_users, rank_, world):
    """Row numbers of ClickData.tsv owned by this rank."""
    return range(rank_, n_users, world)


def allreduce_bucket(flat_grad, group=None, prescaled=True):
    """Sum the dense-gradient bucket over ranks; divide by the world size unless already pre-scaled."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    if not prescaled:
        flat_grad /= world_size(group)
    return flat_grad


def allgather_rows(rows, row_grads, group=None):
    """rows int32[n], row_grads fp32[n, Ue] -> concatenation over ranks in rank order (duplicates kept:
    the optimizer sums them in ascending order, the same on every rank -> replicas stay bit-identical)."""
    w = world_size(group)
    if not (dist.is_available() and dist.is_initialized()):
        return rows, row_grads
    n, ue = row_grads.shape
    all_rows = torch.empty(w * n, dtype=rows.dtype, device=rows.device)
    all_grads = torch.empty(w * n, ue, dtype=row_grads.dtype, device=row_grads.device)
    dist.all_gather_into_tensor(all_rows, rows.contiguous(), group=group)
    dist.all_gather_into_tensor(all_grads, row_grads.contiguous(), group=group)
    return all_rows, all_grads


def allreduce_sums(values, group=None):
    """sum a short list of Python floats over the ranks (float64; evaluation statistics, mnexp_amd/task/paper.py)"""
    t = torch.tensor(list(values), dtype=torch.float64)
    if not (dist.is_available() and dist.is_initialized()) or world_size(group) == 1:
        return t.tolist()
    if dist.get_backend(group) == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().tolist()


def replica_checksum(tensors, group=None):
    """max over ranks of |sum - mean over ranks(sum)| for each tensor: 0.0 when replicas are bit-identical."""
    sums = torch.stack([t.double().sum() for t in tensors])
    if world_size(group) == 1:
        return torch.zeros_like(sums)
    mean = sums.clone()
    dist.all_reduce(mean, group=group)
    mean /= world_size(group)
    dev = (sums - mean).abs()
    dist.all_reduce(dev, op=dist.ReduceOp.MAX, group=group)
    return dev
