"""Data-parallel step glue: one process per GPU, gradients averaged with ONE all-reduce of a flat
bucket per step over RCCL/xGMI (torch.distributed backend 'nccl' is RCCL on ROCm; 'gloo' on CPU for
tests).  Sits between loss.backward() and optimizer.step() (main.py:71-72).  Parameters whose
grad is None (the unused decoder, frozen fine-tune layers) stay out of the bucket and of Adam."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* if WORLD_SIZE > 1; returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def broadcast_parameters(model, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if world_size() == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src)


class GradBucket(object):
    """Flat gradient bucket: pack -> one all-reduce (sum) -> scale by 1/world -> unpack."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.flat = None

    def allreduce_mean(self):
        world = world_size()
        if world == 1:
            return 0
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return 0
        n = sum(g.numel() for g in grads)
        if self.flat is None or self.flat.numel() != n or self.flat.device != grads[0].device:
            self.flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
        views = self.flat.split([g.numel() for g in grads])
        torch._foreach_copy_(list(views), [g.reshape(-1) for g in grads])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / world)
        torch._foreach_copy_([g.view(-1) for g in grads], list(views))
        return n


def shard_range(n_global, rank, world):
    """Contiguous equal shard [lo, hi) of a global batch (global batch must divide evenly)."""
    if n_global % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (n_global, world))
    per = n_global // world
    return rank * per, (rank + 1) * per
