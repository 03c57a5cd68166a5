"""Data-parallel gradient exchange (new work: the reference is single-device).

One process per GPU; each rank runs forward+backward on its own minibatch; ONE exchange step: SUM all-reduce of the
flat fp32 gradient buffer (loss is a plain sum, so SUM -- not mean -- reproduces the N*B-batch gradient exactly), then
every rank applies the same clip + Adam update, so replicas stay identical.  Backend "nccl" is RCCL on ROCm (xGMI);
"gloo" is used by the CPU tests.  The buffer is reduced in a few large buckets issued back-to-back (xGMI rings are
per-link bound, so few large messages beat many small ones)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class GradReducer:
    def __init__(self, bucket_floats=2 * 1024 * 1024, group=None):
        self.bucket = int(bucket_floats)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def all_reduce(self, grads: torch.Tensor, loss: torch.Tensor | None = None):
        if self.world == 1:
            return
        works = []
        n = grads.numel()
        for off in range(0, n, self.bucket):
            works.append(dist.all_reduce(grads[off:min(n, off + self.bucket)], op=dist.ReduceOp.SUM, group=self.group,
                                         async_op=True))
        if loss is not None:
            works.append(dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
