"""One hipGraph per batch shape: the ~850 kernel launches of a training step (forward + backward + clip + AdamW + weight
re-casts; N > 1: + the RCCL collectives) replayed by the GPU front-end instead of being issued one by one from Python.

The reference has nothing like this (its step is ATen eager, lightning.py:86-114); it is the MI355X answer to a step that is
~20 ms of GPU work behind ~25 ms of Python launch overhead.  HIP graphs, not a tracing compiler: the graph holds exactly the
launches the eager step made.  What makes it valid for TRAINING (new data every step) and not just for a benchmark loop:

* the hot path is free of host decisions that depend on tensor VALUES (E2E.forward_tensors: masks, label preparation, CTC
  recursions, loss denominators, dropout seeds, step count, learning rate and clip coefficient all live on the device), so a
  graph captured on one batch is the correct program for every batch of the same SHAPE (B, T, L);
* inputs are copied into the graph's static buffers before each replay; outputs (the loss terms) are static tensors the
  replay rewrites.

A length-bucketed sampler revisits a few hundred (B, T, L) shapes; a shape is captured the second time it shows up (the first
visit runs eagerly and warms every cache the capture must not touch), graphs share one memory pool (they never run
concurrently), and the least recently used graph is dropped beyond `max_graphs`.

Used by bench.py (the benchmark loop), train_native.fit (train.py's native driver) and tests/test_e2e_gpu.py."""
import collections

import torch


class StepGraphs:
    def __init__(self, eager_step, *, enabled=True, capture_after=1, max_graphs=64, thread_local=False, warm=None, on_fail=None):
        """eager_step(x, lens, y) -> tuple of device tensors: the WHOLE step.  It must START by dropping the parameters' gradient
        tensors (`p.grad = None`): the gradients a replay leaves behind live in the graph's memory pool, and an eager step (or a
        later capture) that found them would accumulate into them.  capture_after: eager visits of a shape before it is captured
        (0 = capture on first sight after a warm-up run of `warm(x, lens, y)` on a side stream -- what a benchmark wants).
        thread_local: capture mode for processes whose other threads touch the device (a process group's watchdog).
        on_fail(exc): called when a capture fails; the step then runs eagerly from there on (None: re-raise)."""
        self.eager_step, self.enabled = eager_step, enabled
        self.capture_after, self.max_graphs, self.thread_local = capture_after, max_graphs, thread_local
        self.warm, self.on_fail = warm, on_fail
        self.graphs = collections.OrderedDict()  # shape key -> (graph, static x, static lens, static y, static outputs)
        self.seen = collections.Counter()
        self.pool = None
        self.stats = {"eager": 0, "captured": 0, "replayed": 0, "evicted": 0}

    def reset(self):
        """Drop every captured graph (a change of numerical mode, of the optimizer, of the model).  The shared memory pool goes
        with them: a pool handle whose last graph is gone must not be handed to a new capture."""
        self.graphs.clear()
        self.seen.clear()
        self.pool = None

    @staticmethod
    def key(x, lens, y):
        return (tuple(x.shape), tuple(lens.shape), tuple(y.shape), x.dtype, y.dtype)

    def _capture(self, key, x, lens, y):
        sx, sl, sy = x.clone(), lens.clone(), y.clone()
        if self.warm is not None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.warm(sx, sl, sy)
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        kw = {"capture_error_mode": "thread_local"} if self.thread_local else {}
        with torch.cuda.graph(g, pool=self.pool, **kw):
            outs = self.eager_step(sx, sl, sy)
        outs = tuple(o.detach() for o in (outs if isinstance(outs, (tuple, list)) else (outs,)))
        self.graphs[key] = (g, sx, sl, sy, outs)
        self.stats["captured"] += 1
        while len(self.graphs) > self.max_graphs:
            self.graphs.popitem(last=False)  # (never the one just captured: max_graphs >= 1)
            self.stats["evicted"] += 1

    def __call__(self, x, lens, y):
        """Run one step on (x, lens, y); returns the step's outputs (static tensors of the graph when replayed: read or copy
        them before the next call with the same shape)."""
        if not self.enabled:
            self.stats["eager"] += 1
            return self.eager_step(x, lens, y)
        key = self.key(x, lens, y)
        ent = self.graphs.get(key)
        if ent is None:
            if self.seen[key] < self.capture_after:
                self.seen[key] += 1
                self.stats["eager"] += 1
                return self.eager_step(x, lens, y)
            try:
                self._capture(key, x, lens, y)
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation: the eager step is always valid
                if self.on_fail is None:
                    raise
                self.on_fail(e)
                self.enabled = False
                torch.cuda.synchronize()
                self.stats["eager"] += 1
                return self.eager_step(x, lens, y)
            ent = self.graphs[key]
            g, sx, sl, sy, outs = ent
        else:
            self.graphs.move_to_end(key)
            g, sx, sl, sy, outs = ent
            sx.copy_(x, non_blocking=True)
            sl.copy_(lens, non_blocking=True)
            sy.copy_(y, non_blocking=True)
        g.replay()
        self.stats["replayed"] += 1
        return outs
