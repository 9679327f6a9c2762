"""torch.optim.Adam replacement: every parameter updated by ONE multi-tensor HIP launch per step
(csrc/adam.hip).  Same hyper-parameters, state layout ('step', 'exp_avg', 'exp_avg_sq') and update
formula as torch.optim.Adam (main.py:280), so optimizer checkpoints interchange."""
import ctypes as C

import numpy as np
import torch

from . import ops as _ops
from ._lib import call

_CHUNK = 16384


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._plan_key, self._plan = None, None
        self.graph_tables = None     # ... and per group (pinned host, device) pointer tables [n parameters with a gradient][5]
        self.graph_hyper = None      # graph_step.GraphedStep: per group a device tensor of the six step scalars (cy_adam_multi_dev)

    def step_scalars(self, group, t):
        """{lr, beta1, beta2, eps, 1 - beta1^t, 1 - beta2^t}: the scalars of step t of a group."""
        b1, b2 = group['betas']
        return [float(group['lr']), float(b1), float(b2), float(group['eps']), float(1.0 - b1 ** t), float(1.0 - b2 ** t)]

    def _build_plan(self, entries, device):
        """Pointer table + block map on the device.  The block map depends on the sizes only and is uploaded once; the
        table (the gradient tensors are new allocations every step) goes through a small ring of pinned staging buffers
        with an asynchronous copy on the step's stream: no host synchronisation in the step."""
        blocks = []
        for k, (p, g, m, v) in enumerate(entries):
            blocks.extend((k, c) for c in range((p.numel() + _CHUNK - 1) // _CHUNK))
        sizes = tuple(e[0].numel() for e in entries)
        if getattr(self, '_bm_key', None) != (sizes, device):
            bm = np.asarray(blocks, dtype=np.int32).reshape(-1, 2)
            self._bm, self._bm_key = torch.from_numpy(bm).to(device), (sizes, device)
            self._table_dev = torch.empty((len(entries), 5), dtype=torch.int64, device=device)
            self._ring = [torch.empty((len(entries), 5), dtype=torch.int64).pin_memory() for _ in range(4)]
            self._ring_events = [None] * 4
            self._ring_pos = 0
        k = self._ring_pos
        self._ring_pos = (k + 1) % 4
        if self._ring_events[k] is not None:
            self._ring_events[k].synchronize()           # four steps old: long done
        host = self._ring[k]
        host.numpy()[...] = np.asarray([(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
                                        for p, g, m, v in entries], dtype=np.int64)
        self._table_dev.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._ring_events[k] = ev
        return (self._table_dev, self._bm, len(blocks))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            entries, steps = [], set()
            for p in group['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError('capsyolo_amd.optim.Adam needs contiguous float32 GPU parameters')
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] = int(st['step']) + 1
                steps.add(st['step'])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                entries.append((p, g, st['exp_avg'], st['exp_avg_sq']))
            if not entries:
                continue
            if len(steps) != 1:
                raise RuntimeError('parameters of one group must share their step count')
            t = steps.pop()
            if self.graph_hyper is not None:
                # inside a captured step (graph_step.GraphedStep): the pointer table is a tensor of the graph's own (uploaded by a
                # captured copy from a pinned buffer that nothing else writes), the block map is the eager plan's (same sizes:
                # the warm-up steps built it), the scalars come from device memory
                sizes = tuple(e[0].numel() for e in entries)
                if getattr(self, '_bm_key', None) != (sizes, entries[0][0].device):
                    raise RuntimeError('capsyolo_amd.optim.Adam: capture a step only after an eager step with the same parameters')
                # (pinned and device buffers made by GraphedStep BEFORE the capture: pinning memory is not a capturable operation)
                host, table = self.graph_tables[id(group)]
                if tuple(host.shape) != (len(entries), 5):
                    raise RuntimeError('capsyolo_amd.optim.Adam: the captured step updates %d parameters, its table was made for %d'
                                       % (len(entries), host.shape[0]))
                host.numpy()[...] = np.asarray([(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
                                                for p, g, m, v in entries], dtype=np.int64)
                table.copy_(host, non_blocking=True)
                nblocks = sum((p.numel() + _CHUNK - 1) // _CHUNK for p, _, _, _ in entries)
                call('cy_adam_multi_dev', C.c_void_p(table.data_ptr()), C.c_void_p(self._bm.data_ptr()), nblocks, _CHUNK,
                     C.c_void_p(self.graph_hyper[id(group)].data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                _ops._bump_param_epoch()
                continue
            key = tuple(x.data_ptr() for e in entries for x in e)
            if key != self._plan_key:
                self._plan, self._plan_key = self._build_plan(entries, entries[0][0].device), key
            table, bm, nblocks = self._plan
            b1, b2 = group['betas']
            call('cy_adam_multi', C.c_void_p(table.data_ptr()), C.c_void_p(bm.data_ptr()), nblocks, _CHUNK,
                 float(group['lr']), float(b1), float(b2), float(group['eps']), float(1.0 - b1 ** t),
                 float(1.0 - b2 ** t), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _ops._bump_param_epoch()          # parameters changed through raw pointers: eval-mode fold caches are stale
        return loss
