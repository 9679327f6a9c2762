"""One training step (forward + loss + backward + Adam: main.py:55-80) captured ONCE in a HIP graph and replayed per batch.

Why: the small models are launch-bound -- CapsuleNet's step is ~190 kernel launches that add up to 2.0 ms of kernel time inside a
2.4 ms step (profiles/r04_capsule_kernel_stats.csv); a graph replay issues them without the interpreter and the ctypes calls in
between.  The large models gain nothing (their steps are kernel-bound) and keep the eager loop.

How: every launch of the step goes to torch's current stream (ops.py passes `torch.cuda.current_stream()` to every C-ABI call)
and no kernel allocates, so `torch.cuda.graph` captures the step as it is: hipStreamBeginCapture on torch's capture stream, the
tensors the step creates come from the graph's private pool and keep their addresses across replays.  What changes from step to
step lives in device memory: the batch (static input buffers, filled by `copy_` before the replay) and the optimizer's scalars
(step count -> bias corrections, a scheduler's learning rate: `optim.Adam.graph_hyper`, six floats refreshed before the replay).
Single process only: the data-parallel gradient all-reduce stays on the eager path.
"""
import torch

from . import ops


class GraphedStep(object):
    """step = GraphedStep(model, forward_loss, optimizer, example_batch); loss = step(*batch) per batch.

    forward_loss(model, *batch) -> (y_hat, loss) runs the model and the loss (main.py's `_forward`).  `example_batch`: tensors on
    the device with the shapes / dtypes of every later batch (a batch of another shape falls back to the eager step)."""

    def __init__(self, model, forward_loss, optimizer, example_batch, warmup=3):
        if not torch.cuda.is_available():
            raise ops._lib.HipExtensionError('GraphedStep needs the GPU: there is no CPU path')
        self.model, self.forward_loss, self.opt = model, forward_loss, optimizer
        self.static = [t.clone() for t in example_batch]
        self.y_hat, self.loss, self.graph = None, None, None
        dev = self.static[0].device
        # the scalars travel through a ring of pinned buffers (a buffer is rewritten only after the copy that read it has run: the
        # host may be several replays ahead of the device)
        self._ring = [dict((id(g), torch.empty(6, dtype=torch.float32).pin_memory()) for g in optimizer.param_groups) for _ in range(8)]
        self._ring_events, self._ring_pos = [None] * 8, 0
        self._hyper_dev = dict((id(g), torch.zeros(6, dtype=torch.float32, device=dev)) for g in optimizer.param_groups)
        # eager warm-up steps on a side stream (torch's capture recipe): optimizer state, the zero pool and the kernels' one-time
        # attribute calls (cy_allow_lds) exist before the capture begins.  They must not train: parameters, buffers (BatchNorm running
        # statistics) and optimizer state are put back IN PLACE afterwards (the capture holds their addresses).
        snap_model = dict((k, v.detach().clone()) for k, v in model.state_dict().items())
        snap_opt = dict((p, dict((k, (v.detach().clone() if torch.is_tensor(v) else v)) for k, v in optimizer.state[p].items()))
                        for g in optimizer.param_groups for p in g['params'] if p in optimizer.state and len(optimizer.state[p]))
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager(*self.static)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(snap_model[k])
            for g in optimizer.param_groups:
                for p in g['params']:
                    st = optimizer.state.get(p)
                    if not st:
                        continue
                    old = snap_opt.get(p)
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            v.copy_(old[k]) if old is not None else v.zero_()
                        else:
                            st[k] = old[k] if old is not None else 0
        ops._bump_param_epoch()
        self.graph = torch.cuda.CUDAGraph()
        self._refresh_scalars(advance=False)          # valid numbers during the capture (they are not read until a replay)
        optimizer.graph_hyper = self._hyper_dev
        # the optimizer's pointer table of the captured step: as many rows as parameters that got a gradient in the warm-up steps
        self._tables = {}
        for g in optimizer.param_groups:
            n = sum(1 for q in g['params'] if q.grad is not None)
            self._tables[id(g)] = (torch.empty((n, 5), dtype=torch.int64).pin_memory(), torch.empty((n, 5), dtype=torch.int64, device=dev))
        optimizer.graph_tables = self._tables
        try:
            optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self.y_hat, self.loss = forward_loss(model, *self.static)
                self.loss.backward()
                self._captured_step()
        finally:
            optimizer.graph_hyper = None
            optimizer.graph_tables = None

    def _eager(self, *batch):
        y_hat, loss = self.forward_loss(self.model, *batch)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return y_hat, loss

    def _captured_step(self):
        # optim.Adam.step with graph_hyper set launches cy_adam_multi_dev; its per-parameter step counters must not move during the
        # capture (the replay's counters are advanced by _refresh_scalars)
        saved = dict((p, int(self.opt.state[p]['step'])) for g in self.opt.param_groups for p in g['params'] if p in self.opt.state and len(self.opt.state[p]))
        self.opt.step()
        for p, t in saved.items():
            self.opt.state[p]['step'] = t

    def _refresh_scalars(self, advance=True):
        k = self._ring_pos
        self._ring_pos = (k + 1) % len(self._ring)
        if self._ring_events[k] is not None:
            self._ring_events[k].synchronize()
        for g in self.opt.param_groups:
            ps = [p for p in g['params'] if p in self.opt.state and len(self.opt.state[p])]
            if not ps:
                continue
            t = int(self.opt.state[ps[0]]['step']) + (1 if advance else 0)
            if advance:
                for p in ps:
                    self.opt.state[p]['step'] = t
            host = self._ring[k][id(g)]
            host.copy_(torch.tensor(self.opt.step_scalars(g, max(t, 1)), dtype=torch.float32))
            self._hyper_dev[id(g)].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.static[0].device))
        self._ring_events[k] = ev

    def __call__(self, *batch):
        """One training step on `batch`; returns (y_hat, loss) -- the graph's static output tensors (overwritten by the next call)."""
        if any(b.shape != s.shape or b.dtype != s.dtype for b, s in zip(batch, self.static)):
            return self._eager(*batch)                # a ragged last batch: the eager step (same kernels)
        for s, b in zip(self.static, batch):
            if s.data_ptr() != b.data_ptr():
                s.copy_(b, non_blocking=True)
        self._refresh_scalars(advance=True)
        self.graph.replay()
        ops._bump_param_epoch()
        return self.y_hat, self.loss
