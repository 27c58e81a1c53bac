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

A length-bucketed sampler revisits a few hundred (B, T, L) shapes (384 at max-frames 1600 on the synthetic corpus); a shape is
captured the second time it shows up (the first visit runs eagerly and warms every cache the capture must not touch) and graphs
share one memory pool (they never run concurrently).  Capacity policy (round 6): once `max_graphs` shapes are captured, further
shapes simply run eagerly -- NO eviction by default.  An LRU that evicts on every miss thrashes as soon as the shape count
exceeds the capacity (capture + replay costs more than an eager step, and every capture pins per-graph resources of the
optimizer and of the gradient buckets); `evict=True` restores it for callers that really cycle through working sets, and then
`on_evict(key)` lets the owners of per-capture resources take them back (optim.FusedAdamW.release_captured,
ddp.GradBuckets.release_captured -- keyed by `capture_token()`, the shape key of the capture in progress).  The default capacity
of train.py's loop is 512 graphs: with 288 GB of HBM the ~50 MB of static inputs per shape is not a constraint.

Used by bench.py (the benchmark loop), train_native.fit (train.py's native driver) and tests/test_e2e_gpu.py."""
import collections

import torch

_capture_token = None


def capture_token():
    """Shape key of the StepGraphs capture in progress (None outside one): what per-capture resources are tagged with."""
    return _capture_token


class StepGraphs:
    def __init__(self, eager_step, *, enabled=True, capture_after=1, max_graphs=64, thread_local=False, warm=None, on_fail=None,
                 evict=False, on_evict=None):
        """eager_step(x, lens, y) -> tuple of device tensors: the WHOLE step.  It must START by dropping the parameters' gradient
        tensors (`p.grad = None`): the gradients a replay leaves behind live in the graph's memory pool, and an eager step (or a
        later capture) that found them would accumulate into them.  capture_after: eager visits of a shape before it is captured
        (0 = capture on first sight after a warm-up run of `warm(x, lens, y)` on a side stream -- what a benchmark wants).
        thread_local: capture mode for processes whose other threads touch the device (a process group's watchdog).
        on_fail(exc): called when a capture fails; the step then runs eagerly from there on (None: re-raise).
        evict: drop the least recently used graph beyond max_graphs (default: keep the first max_graphs shapes, run the rest
        eagerly); on_evict(key): called for every dropped graph."""
        self.eager_step, self.enabled = eager_step, enabled
        self.evict, self.on_evict = evict, on_evict
        self.capture_after, self.max_graphs, self.thread_local = capture_after, max_graphs, thread_local
        self.warm, self.on_fail = warm, on_fail
        self.graphs = collections.OrderedDict()  # shape key -> (graph, static x, static lens, static y, static outputs)
        self.seen = collections.Counter()
        self.pool = None
        self.stats = {"eager": 0, "captured": 0, "replayed": 0, "evicted": 0, "full": 0}

    def reset(self):
        """Drop every captured graph (a change of numerical mode, of the optimizer, of the model).  The shared memory pool goes
        with them: a pool handle whose last graph is gone must not be handed to a new capture."""
        if self.on_evict is not None:
            for k in list(self.graphs):
                self.on_evict(k)
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
        global _capture_token
        _capture_token = key
        try:
            with torch.cuda.graph(g, pool=self.pool, **kw):
                outs = self.eager_step(sx, sl, sy)
        finally:
            _capture_token = None
        outs = tuple(o.detach() for o in (outs if isinstance(outs, (tuple, list)) else (outs,)))
        self.graphs[key] = (g, sx, sl, sy, outs)
        self.stats["captured"] += 1
        while len(self.graphs) > self.max_graphs:
            old, _ = self.graphs.popitem(last=False)  # (never the one just captured: max_graphs >= 1)
            self.stats["evicted"] += 1
            if self.on_evict is not None:
                self.on_evict(old)

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
            if not self.evict and len(self.graphs) >= self.max_graphs:
                self.stats["full"] += 1  # at capacity: this shape stays eager (no LRU thrash, no per-capture resources spent)
                self.stats["eager"] += 1
                return self.eager_step(x, lens, y)
            try:
                self._capture(key, x, lens, y)
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation: the eager step is always valid
                if self.on_fail is None:
                    raise
                self.on_fail(e)
                self.enabled = False
                if self.on_evict is not None:
                    self.on_evict(key)  # whatever the half-finished capture took
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
