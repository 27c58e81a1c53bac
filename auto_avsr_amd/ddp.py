"""Data-parallel gradient exchange of the native training loop (train.py:30-42: `DDPStrategy(find_unused_parameters=False)` --
every rank holds a full replica, gradients are averaged over the ranks once per step), written directly on RCCL collectives
instead of `torch.nn.parallel.DistributedDataParallel`:

* parameters are assigned, in reverse registration order (roughly the order their gradients become ready in the backward pass),
  to flat f32 **buckets** of `bucket_mb` MB -- large messages, because a ring all-reduce over point-to-point xGMI links is
  bound per link (7 x ~153 GB/s per GPU);
* a post-accumulate-grad hook per parameter counts a bucket down; when its last gradient has arrived, ONE launch
  (`avsr_multi_copy_scale`) gathers the bucket's gradients into the flat buffer, pre-divided by the world size, the parameters'
  `.grad` are re-pointed at their slices of it, and `all_reduce(async_op=True)` goes out on RCCL's own stream -- overlapped
  with the rest of the backward pass (the parameter-poor, compute-rich ResNet trunk runs last);
* `begin_step()` (before the forward pass, on the thread and stream that issue the step) records the compute stream; every
  gather is then issued ON that stream from the hook -- not on the AccumulateGrad node's own stream, which need not be the
  producer's (the allocator and, in a replayed graph, the ordering hazards of a foreign-stream launch: see `_flush`);
* `finish()` makes the compute stream wait for the outstanding reductions (before the optimizer reads the gradients).

Two transports.  `group=` : torch.distributed (`all_reduce(async_op=True)`, any backend -- what the CPU suite runs on gloo).
`comm=` : a `comm.StreamComm` -- RCCL's C API inside libavsr_hip.so, the all-reduce forked to a side stream of this object
behind an event: plain stream operations, so the WHOLE data-parallel step (buckets, cross-rank BatchNorm through a second
communicator, optimizer) can be captured into a hipGraph and replayed (`bench.py --ddp auto`).  torch DDP's reducer cannot be
captured on this stack (`tools/rccl_capture_probe.py`), and torch collectives inside a capture leave Work objects whose events
the process-group watchdog polls.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import ops

_CHUNK = 4096  # elements per block of avsr_multi_copy_scale (csrc/optim.hip OPT_CHUNK)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class GradBuckets:
    def __init__(self, params, group=None, bucket_mb=None, comm=None, wire=None, spare=72, shard=False):
        """shard (round 6, optim.ShardedAdamW): the exchange is a REDUCE-SCATTER -- after it a rank holds the averaged gradients of
        its 1 / world slice of every bucket only -- and the parameters themselves live in flat per-bucket buffers laid out like
        the gradient buckets (`pflat`; `p.data` re-pointed at its slice), so that the sharded optimizer can update a contiguous
        slice and all-gather the buffer.  Half the wire bytes of the all-reduce inside the backward pass; the other half is the
        all-gather of the updated weights after the optimizer.
        spare: captured step shapes this object can serve (one pinned pointer table per bucket and shape); size it to the
        owner's graph capacity (train_native: StepGraphs max_graphs + 8).
        bucket_mb: bucket size in MB of f32 gradients (default 64, environment AVSR_BUCKET_MB overrides: sweep hook).
        wire: "f32" (default) or "bf16" (environment AVSR_GRAD_WIRE when the argument is None): the format the buckets travel in.  bf16 halves the bytes
        per xGMI link -- the exchange of 1.0 GB of f32 gradients is per-link bound on a ring -- at the
        price of bf16 sums across the ranks (8 significant bits; the bf16 / mixed modes compute their gradients from bf16
        operands anyway); the reduced bucket is widened back to f32 for the optimizer on the exchange's own stream."""
        import os

        if bucket_mb is None:
            bucket_mb = float(os.environ.get("AVSR_BUCKET_MB", "64"))
        # an explicit argument wins; the environment only replaces the default (sweep hook).  The default is f32 -- the reference's
        # all-reduce (train.py:30-42: torch DDP sums f32 gradients); bf16 is an opt-in (round-5 advisor finding)
        self.wire = wire or os.environ.get("AVSR_GRAD_WIRE") or "f32"
        assert self.wire in ("f32", "bf16"), self.wire
        if self.wire == "bf16" and comm is None:
            self.wire = "f32"  # the narrow format needs a stream communicator; torch.distributed groups travel as f32
        self.timing = None  # set by time_next_step(): per-bucket events of ONE eager step (diagnostics)
        self.params = [p for p in params if p.requires_grad]
        assert self.params and all(p.dtype == torch.float32 for p in self.params)
        self.group = group
        # comm: a comm.StreamComm over the same ranks -- the bucket all-reduces then go straight to RCCL on a side stream of
        # this object (forked from / joined to the compute stream with events: capturable), not through torch.distributed
        self.comm = comm
        self.world = comm.world if comm is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.device = self.params[0].device
        self._cap = max(1, int(bucket_mb * (1 << 20) / 4))
        self._n_spare = spare
        self.shard = bool(shard)
        self.rank = comm.rank if comm is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        if self.shard:
            assert self.wire == "f32", "sharded optimizer: f32 wire"
        # first assignment: the parameters in reverse registration order (roughly the order their gradients become ready); the
        # first step records the REAL arrival order and rebuild_by_arrival() re-buckets by it (see there)
        self._arrival = []
        self.rebuilt = False
        self._assign(list(reversed(range(len(self.params)))))
        if self.shard:
            # flat parameter buffers, same offsets as the gradient buckets; the parameters become views of them (their identity,
            # hence state_dict / autograd, is untouched; cached operand copies keyed by address are not: the caller invalidates)
            self.rebuilt = True  # (the layout is fixed from here on: no re-bucketing by arrival order)
            self._arrival = None
            self.pflat = [torch.zeros_like(f) for f in self.flat]
            with torch.no_grad():
                for i, p in enumerate(self.params):
                    b, o = self.bucket_of[i], self.offset[i]
                    v = self.pflat[b][o:o + p.numel()].view_as(p)
                    v.copy_(p.data)
                    p.data = v
        self.compute_stream = None  # set by begin_step(): the stream the step's kernels are issued on
        self._side = torch.cuda.Stream(device=self.device) if (comm is not None and self.device.type == "cuda") else None
        self._side_used = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]

    def _assign(self, order):
        """Buckets of at most `_cap` f32 elements over the parameters in `order`; every slice starts 16-byte aligned."""
        self.bucket_of, self.offset, sizes, members = {}, {}, [], []
        cur, cur_n = [], 0
        for i in order:
            n = (self.params[i].numel() + 3) // 4 * 4
            if cur and cur_n + n > self._cap:
                members.append(cur)
                sizes.append(cur_n)
                cur, cur_n = [], 0
            self.bucket_of[i], self.offset[i] = len(members), cur_n
            cur.append(i)
            cur_n += n
        members.append(cur)
        sizes.append(cur_n)
        if getattr(self, "shard", False):  # every rank's slice of a bucket starts 16-byte aligned
            q = 4 * self.world
            sizes = [(n + q - 1) // q * q for n in sizes]
        self.members = members
        self.flat = [torch.zeros(n, dtype=torch.float32, device=self.device) for n in sizes]
        self.narrow = [torch.zeros(n, dtype=torch.bfloat16, device=self.device) for n in sizes] if self.wire == "bf16" else None
        self.views = {i: self.flat[self.bucket_of[i]][self.offset[i]: self.offset[i] + self.params[i].numel()].view_as(self.params[i])
                      for i in range(len(self.params))}
        self._left = [len(m) for m in members]
        self._works = []
        # Pointer tables of the gather launches.  A captured step keeps ONE (pinned host rows, device table) pair per bucket for
        # good -- the capture records a copy node that reads the pinned rows on every replay.  Eager steps (gradient addresses
        # change from step to step) rotate through a small ring per bucket whose slots carry an event: a slot's pinned rows are
        # rewritten only after the asynchronous copy that last read them has completed.  Everything pinned is allocated HERE
        # (hipHostMalloc is not allowed under stream capture).
        self._captured = {}  # (bucket, gradient addresses) -> (host rows, device table, blocks)
        self._spare = [[self._new_slot(len(m)) for _ in range(self._n_spare)] for m in members]  # for captured steps (one per batch shape)
        self._ring = [[self._new_slot(len(m)) for _ in range(4)] for m in members]    # for eager steps
        self._ring_pos = [0] * len(members)
        self._seen = [dict() for _ in members]  # per bucket: the streams its parameters' hooks ran on in this step

    def rebuild_by_arrival(self):
        """Re-bucket the parameters in the order their gradients ARRIVED during the first step (what torch DDP does after its
        first iteration).  Registration order is a poor predictor here: one autograd node produces the `linear_pos` weight
        gradients of ALL twelve encoder layers at the very end of the encoder's backward pass, so under the first assignment
        every encoder bucket holds one late member and all of them flush together, ten buckets at once with only the trunk's
        backward left to hide under (round 5, `config.bucket_overlap` of a one-rank run: encoder buckets at 35.3 - 36.6 ms of
        a step whose decoder buckets left at 22.5 ms).  In arrival order the late gradients share the LAST bucket and every
        other bucket leaves as soon as its layers are done.  Call once, after the first (eager) step, with the gradients
        cleared, before any hipGraph capture; every rank must call it (rank 0's order is broadcast: one order for all)."""
        assert not self._captured and all(n == len(m) for n, m in zip(self._left, self.members)), "rebuild between steps, before captures"
        order = list(self._arrival) if self._arrival is not None else []  # (None: finish() dropped a record older than one step)
        self._arrival = None
        self.rebuilt = True
        valid = sorted(order) == list(range(len(self.params)))  # a step that touched every parameter exactly once
        if self.world > 1 and self.group is not None and dist.is_initialized():
            # ONE decision for all ranks, taken collectively BEFORE anybody may leave: rank 0's order (all -1 when rank 0 has no
            # valid record) is broadcast; a rank whose own record is unusable still takes part and adopts rank 0's
            t = torch.tensor(order if valid else [-1] * len(self.params), dtype=torch.int64,
                             device=self.device if dist.get_backend(self.group) != "gloo" else "cpu")
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not dist.group.WORLD else 0, group=self.group)
            order = t.tolist()
            valid = sorted(order) == list(range(len(self.params)))
        if not valid:
            return False  # keep the first assignment (every rank decides the same way)
        for p in self.params:
            assert p.grad is None, "rebuild_by_arrival: clear the gradients first (they are views of the old buckets)"
        self._assign(order)
        return True

    def _new_slot(self, n):
        host = torch.empty(48 * n, dtype=torch.uint8)
        if self.device.type == "cuda":
            host = host.pin_memory()
        return [host, torch.empty(48 * n, dtype=torch.uint8, device=self.device), None]  # rows, device table, event of the last copy

    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            if self._arrival is not None:
                self._arrival.append(i)
            if self.device.type == "cuda":
                # a post-accumulate hook runs on the stream of its parameter's AccumulateGrad node -- not necessarily the
                # same stream for every parameter of the bucket (nodes that survived from an earlier iteration keep theirs)
                st = torch.cuda.current_stream()
                self._seen[b][st.cuda_stream] = st
                if self.compute_stream is not None and st != self.compute_stream:
                    # (round 6) a branch of the step that ran on a second stream (E2E's CTC branch): its gradients were produced
                    # there, the gather below is issued on the compute stream -- order the two now (the bucket may complete much
                    # later, from the compute stream)
                    self.compute_stream.wait_stream(st)
            self._left[b] -= 1
            if self._left[b] == 0:
                self._flush(b)
        return hook

    def _flush(self, b):
        idx = self.members[b]
        grads = [self.params[i].grad for i in idx]
        assert all(g is not None and g.dtype == torch.float32 and g.is_contiguous() for g in grads)
        ptrs = tuple(g.data_ptr() for g in grads)
        numel = np.array([self.params[i].numel() for i in idx], dtype=np.int64)
        nblk = (numel + _CHUNK - 1) // _CHUNK
        blocks = int(nblk.sum())

        def fill(host):
            rows = np.zeros((len(idx), 6), dtype=np.uint64)
            rows[:, 0] = [self.views[i].data_ptr() for i in idx]
            rows[:, 1] = ptrs
            if self.narrow is not None and self.comm is not None:
                # bf16 wire: the gather writes the scaled gradients straight into the bf16 bucket (csrc/optim.hip
                # multi_copy_scale_kernel, entry field m); the f32 views are filled by the widening pass after the all-reduce
                rows[:, 2] = [self.narrow[b].data_ptr() + 2 * self.offset[i] for i in idx]
            rows[:, 4] = numel.astype(np.uint64)
            rows[:, 5] = (np.cumsum(nblk) - nblk).astype(np.uint64)
            host.numpy()[:] = rows.reshape(-1).view(np.uint8)

        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        slot = None
        if capturing:
            ent = self._captured.get((b,) + ptrs)
            if ent is None:
                if not self._spare[b]:
                    raise RuntimeError("GradBuckets: out of pre-pinned table buffers under hipGraph capture (`spare` captured step shapes per bucket)")
                from .graph_step import capture_token

                host, dev, _ = self._spare[b].pop()
                fill(host)
                ent = self._captured[(b,) + ptrs] = (host, dev, blocks, capture_token())
            host, dev = ent[0], ent[1]
        else:
            slot = self._ring[b][self._ring_pos[b]]
            self._ring_pos[b] = (self._ring_pos[b] + 1) % len(self._ring[b])
            if slot[2] is not None:
                slot[2].synchronize()  # the copy that last read these pinned rows (four steps ago: long done)
            host, dev = slot[0], slot[1]
            fill(host)
        # Which stream issues the gather.  This hook runs on the stream of the last parameter's AccumulateGrad node, which need
        # not be the stream that produced (and allocated) the gradients: autograd orders the two with events, but (a) the hooks
        # of one bucket may have run on several streams, (b) the caching allocator knows nothing of a launch on a foreign
        # stream and may hand a gradient's block to the producer stream's next allocation before the gather has read it.  In a
        # replayed hipGraph nothing else orders such pairs and the races showed as NaN gradients.  After begin_step() the
        # gather is therefore issued on the COMPUTE stream itself -- the producers' stream; this (autograd) thread issues the
        # backward kernels on it one after the other, so program order is stream order and no allocator hazard exists.
        issue = self.compute_stream if self.compute_stream is not None else (
            torch.cuda.current_stream() if self.device.type == "cuda" else None)
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream()
            if self.compute_stream is None:
                for sid, st in self._seen[b].items():
                    if sid != cur.cuda_stream:
                        cur.wait_stream(st)  # everything the other hooks' streams have accumulated so far
                for g in grads:
                    g.record_stream(cur)
            self._seen[b].clear()
        ctx = torch.cuda.stream(issue) if issue is not None else _NullCtx()
        with ctx:
            dev.copy_(host, non_blocking=True)  # (under capture: a memcpy node reading this pinned buffer on every replay)
            if slot is not None and self.device.type == "cuda":
                slot[2] = torch.cuda.Event()
                slot[2].record()
            ops.call("avsr_multi_copy_scale", ops._ptr(dev), len(idx), blocks, 1.0 / self.world, ops._stream(dev),
                     nbytes=8.0 * self.flat[b].numel())
            for i in idx:
                self.params[i].grad = self.views[i]
            if self.comm is not None:
                # RCCL's C API on the side stream, behind the gather launch; the compute stream goes on with the backward pass
                if self._side is not None:
                    self._side.wait_stream(torch.cuda.current_stream())
                with (torch.cuda.stream(self._side) if self._side is not None else _NullCtx()):
                    if self.timing is not None and self._side is not None:
                        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        ev[0].record()
                        self._reduce(b)
                        ev[1].record()
                        self.timing["buckets"].append((b, ev))
                    else:
                        self._reduce(b)
                self._side_used = self._side is not None
            elif self.world > 1 or self.group is not None:
                # (torch.distributed transport: gloo has no reduce-scatter -- in shard mode the all-reduce, of which the optimizer
                # reads this rank's slice only)
                self._works.append(dist.all_reduce(self.flat[b], group=self.group, async_op=True))

    def shard_range(self, b):
        n = self.flat[b].numel() // self.world
        return self.rank * n, (self.rank + 1) * n

    def _reduce(self, b):
        """The all-reduce of bucket b on the current stream, in the wire format (shard mode: the reduce-scatter)."""
        if self.shard:
            self.comm.reduce_scatter(self.flat[b])
            return
        if self.wire == "bf16":
            # (the bucket is already bf16: the gather launch wrote it -- no pass over an f32 image of the local gradients)
            self.comm.all_reduce(self.narrow[b])
            ops.cast_into(self.narrow[b], self.flat[b])   # back to the f32 views the optimizer reads
        else:
            self.comm.all_reduce(self.flat[b])

    def abort_step(self):
        """A step died between begin_step() and finish() (an exception in the backward pass, a hipGraph capture that failed
        half way): forget its partial bucket counts, pending reductions and stream bookkeeping, so that the next step starts
        clean -- otherwise the buckets that had already counted some gradients in would never flush again and finish() would
        raise on this rank while the others wait in a collective.  The caller clears the parameters' gradients (they may be
        views of the buckets or live in a dead graph's pool)."""
        self._left = [len(m) for m in self.members]
        self._works.clear()
        self._side_used = False
        for s in self._seen:
            s.clear()
        if self._arrival is not None and not self.rebuilt:
            self._arrival = []
        if self._side is not None and self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_stream(self._side)  # nothing of the dead step may still be queued behind us

    def release_captured(self, token):
        """The hipGraph captured under graph_step.capture_token() == token is gone: its pinned pointer tables return to the
        spare lists (StepGraphs on_evict)."""
        if token is None:
            return
        for k in [k for k, e in self._captured.items() if len(e) > 3 and e[3] == token]:
            host, dev = self._captured.pop(k)[:2]
            self._spare[k[0]].append([host, dev, None])

    def spare_left(self):
        return min(len(s) for s in self._spare) if self._spare else 0

    def begin_step(self):
        """Before the forward pass, on the thread / stream that issues the step: remembers the compute stream, so that every
        bucket's gather is ordered behind it explicitly (see _flush)."""
        if self.device.type == "cuda":
            self.compute_stream = torch.cuda.current_stream()

    def finish(self):
        """After loss.backward(): every bucket has been flushed; the compute stream waits for the reductions."""
        missing = [b for b, n in enumerate(self._left) if n != 0]
        if missing:
            raise RuntimeError(f"GradBuckets.finish(): buckets {missing} did not receive all of their gradients "
                               "(a parameter without gradient -- find_unused_parameters=False semantics)")
        for w in self._works:
            w.wait()
        self._works.clear()
        if self._side_used:
            if self.timing is not None:
                self.timing["bwd_end"] = torch.cuda.Event(enable_timing=True)
                self.timing["bwd_end"].record()
            torch.cuda.current_stream().wait_stream(self._side)
            if self.timing is not None:
                self.timing["joined"] = torch.cuda.Event(enable_timing=True)
                self.timing["joined"].record()
            self._side_used = False
        self._left = [len(m) for m in self.members]
        if self._arrival is not None and len(self._arrival) > len(self.params):
            self._arrival = None  # (more than one step went by without a rebuild: the record is of no use any more)

    def time_next_step(self):
        """Diagnostics: the next EAGER step records an event pair around every bucket's exchange on the side stream and around
        the join at the end of the backward pass.  Stream communicators on a GPU only."""
        self.timing = {"buckets": [], "t0": None}
        if self.device.type == "cuda":
            self.timing["t0"] = torch.cuda.Event(enable_timing=True)
            self.timing["t0"].record()

    def timing_report(self):
        """After time_next_step() + one eager step + a device synchronisation: per bucket (in flush order) its size, when its
        exchange started and ended relative to the start of the step, and how long the compute stream stood waiting at the
        join -- the part of the gradient exchange that the backward pass did NOT hide."""
        t, self.timing = self.timing, None
        if not t or t.get("t0") is None or "joined" not in t:
            return None
        t0 = t["t0"]
        rows = [{"bucket": b, "mb": round(self.flat[b].numel() * (2 if self.wire == "bf16" else 4) / 2 ** 20, 1),
                 "start_ms": round(t0.elapsed_time(ev[0]), 3), "end_ms": round(t0.elapsed_time(ev[1]), 3)} for b, ev in t["buckets"]]
        return {"wire": self.wire, "buckets": rows, "backward_done_ms": round(t0.elapsed_time(t["bwd_end"]), 3),
                "exchange_done_ms": round(t0.elapsed_time(t["joined"]), 3),
                "exposed_ms": round(t["bwd_end"].elapsed_time(t["joined"]), 3),
                "sum_exchange_ms": round(sum(r["end_ms"] - r["start_ms"] for r in rows), 3)}

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
