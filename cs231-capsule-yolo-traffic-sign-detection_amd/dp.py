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
        if backend is None:       # CAPSYOLO_DP_BACKEND=gloo: rehearse the N>1 path on fewer GPUs than ranks
            backend = os.environ.get('CAPSYOLO_DP_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_device_index(local_rank, backend))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def local_device_index(local_rank, backend='nccl'):
    """The GPU of this rank.  RCCL wants one GPU per rank: a local rank without a device of its own is an error there.
    Under 'gloo' (rehearsals of the N>1 code path on a box with fewer GPUs than ranks, tests) ranks may share a device."""
    n = torch.cuda.device_count()
    if n == 0 or local_rank < n:
        return local_rank
    if backend == 'nccl':
        raise RuntimeError('local rank %d has no GPU of its own (%d visible): backend nccl (RCCL) needs one GPU per rank; '
                           'use backend gloo to rehearse on shared devices' % (local_rank, n))
    return local_rank % n


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def broadcast_parameters(model, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if world_size() == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src)


class GradBucket(object):
    """Flat gradient bucket between loss.backward() and optimizer.step() (main.py:71-72): every gradient is packed
    into ONE flat buffer pre-scaled by 1 / world (cy_multi_copy: one launch), the buffer is all-reduced (SUM) once, and
    the parameters' .grad become views INTO the flat buffer -- there is no unpack pass, the fused Adam reads the
    averaged gradients where the collective left them.  CPU tensors (gloo tests, the plain-torch `cnn` baseline) take
    torch's foreach ops for the pack."""

    _CHUNK = 16384

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.flat = None
        self._key, self._plan = None, None

    def _pack(self, grads, scale):
        n = sum(g.numel() for g in grads)
        dev = grads[0].device
        if self.flat is None or self.flat.numel() != n or self.flat.device != dev:
            self.flat = torch.empty(n, dtype=grads[0].dtype, device=dev)
        views = self.flat.split([g.numel() for g in grads])
        if not self.flat.is_cuda:
            torch._foreach_copy_(list(views), [g.reshape(-1) for g in grads])
            if scale != 1.0:
                self.flat.mul_(scale)
            return views
        import ctypes as C
        import numpy as np
        from ._lib import call
        grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
        key = tuple(g.data_ptr() for g in grads) + (self.flat.data_ptr(),)
        if key != self._key:
            table = np.zeros((len(grads), 3), dtype=np.int64)
            blocks, off = [], 0
            for k, g in enumerate(grads):
                table[k] = (g.data_ptr(), off, g.numel())
                blocks.extend((k, c) for c in range((g.numel() + self._CHUNK - 1) // self._CHUNK))
                off += g.numel()
            bm = np.asarray(blocks, dtype=np.int32).reshape(-1, 2)
            self._plan = (torch.from_numpy(table).to(dev), torch.from_numpy(bm).to(dev), len(blocks))
            self._key = key
        table, bm, nblocks = self._plan
        call('cy_multi_copy', C.c_void_p(table.data_ptr()), C.c_void_p(bm.data_ptr()), nblocks, self._CHUNK,
             C.c_void_p(self.flat.data_ptr()), 0, float(scale), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return views

    def allreduce_mean(self, force=False):
        """Returns the number of gradient elements that went through the collective (0: single process, nothing done).
        `force` runs the pack -> all-reduce -> re-point path on a world of one (the nccl self-test)."""
        world = world_size()
        if world == 1 and not force:
            return 0
        with_grad = [p for p in self.params if p.grad is not None]
        if not with_grad:
            return 0
        grads = [p.grad for p in with_grad]
        views = self._pack(grads, 1.0 / world)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        for p, v in zip(with_grad, views):
            p.grad = v.view(p.shape)
        return self.flat.numel()


def shard_range(n_global, rank, world):
    """Contiguous equal shard [lo, hi) of a global batch (global batch must divide evenly)."""
    if n_global % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (n_global, world))
    per = n_global // world
    return rank * per, (rank + 1) * per
