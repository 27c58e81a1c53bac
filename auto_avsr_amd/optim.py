"""Fused optimizer step for the native training loop: global-norm gradient clipping, AdamW and the per-step warm-up
cosine learning-rate schedule of the reference (lightning.py:48-52: AdamW(lr, betas=(0.9, 0.98), weight_decay);
train.py:41: gradient_clip_val=10.0; cosine.py:6-25) as THREE kernel launches over all parameters
(csrc/optim.hip: avsr_adamw_step), with the step count, learning rate, gradient norm and clip coefficient resident on the
device -- no host synchronisation, capturable in a hipGraph.

cast_weights=True additionally rewrites the bf16 operand copies of the Linear weights (functional.py weight cache) inside
the update pass (avsr_adamw_cast_step): the step then leaves the model ready for the next forward pass, and the separate
re-cast launch -- a second read of every f32 weight -- disappears from the training step."""
import numpy as np
import torch

from . import ops

_CHUNK = 4096


class FusedAdamW:
    def __init__(self, params, lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0, warmup_steps=0,
                 total_steps=0, cast_weights=False, graph_shapes=64):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        assert all(p.dtype == torch.float32 and p.is_contiguous() for p in self.params), "f32 contiguous master weights"
        self.device = self.params[0].device
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.max_grad_norm, self.warmup_steps, self.total_steps = float(max_grad_norm), int(warmup_steps), int(total_steps)
        if self.total_steps > 0 and self.total_steps <= self.warmup_steps:
            # (the reference's scheduler divides by total - warm-up steps, cosine.py:23: a ZeroDivisionError there -- here the
            # device-side schedule would turn the learning rate, and with it every weight, into NaN without a word)
            raise ValueError(f"FusedAdamW: total_steps ({self.total_steps}) must exceed warmup_steps ({self.warmup_steps})")
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.state = torch.zeros(4, dtype=torch.float32, device=self.device)  # step, lr, grad norm, clip coefficient
        self._tables = {}  # gradient addresses -> (pinned host table, device table, n, blocks, scratch, made under capture)
        # pinned staging buffers are allocated HERE: hipHostMalloc is not allowed while a stream is capturing, and the
        # pointer table of a captured step can only be built during the capture (that is when its gradients exist)
        self.cast_weights = bool(cast_weights)
        self._cast_plan = None  # (generation, covered parameter mask, linear rows, linear blocks, tile rows, tile blocks)
        self._table_bytes = (48 + 48 + 80) * len(self.params)  # full + linear + tile tables share one buffer
        # table rows {p, g, m, v, numel, blk0 | 0 << 32} as six u64 (csrc/optim.hip OptEntry): only column 1 (the gradient
        # address) changes between steps
        n = len(self.params)
        self._rows = np.zeros((n, 6), dtype=np.uint64)
        self._rows[:, 0] = [p.data_ptr() for p in self.params]
        self._rows[:, 2] = [m.data_ptr() for m in self.exp_avg]
        self._rows[:, 3] = [v.data_ptr() for v in self.exp_avg_sq]
        numel = np.array([p.numel() for p in self.params], dtype=np.int64)
        blocks = (numel + _CHUNK - 1) // _CHUNK
        self._rows[:, 4] = numel.astype(np.uint64)
        self._rows[:, 5] = (np.cumsum(blocks) - blocks).astype(np.uint64)
        self._blocks = int(blocks.sum())
        # pinned table buffers: every captured step shape holds one for good (hipHostMalloc is not allowed under capture, so they
        # are all allocated here), eager steps rotate through up to 16 more (gradient addresses change from step to step)
        self._free_host = [self._new_host() for _ in range(graph_shapes + 24)]

    def _new_host(self):
        t = torch.empty(self._table_bytes, dtype=torch.uint8)
        return t.pin_memory() if self.device.type == "cuda" else t

    # -- gradient pointer table: gradients are fresh tensors after every backward, so their addresses may move
    def _plan(self):
        """Split of the parameters into tile-updated Linear weights (with their bf16 copies) and the rest, for the
        current content of the functional weight cache.  None: nothing registered (yet) -- plain step."""
        from . import functional as AF

        gen, groups = AF.weight_cast_groups()
        if self._cast_plan is not None and self._cast_plan[0] == gen:
            return self._cast_plan
        index = {p.data_ptr(): i for i, p in enumerate(self.params)}
        covered = np.zeros(len(self.params), dtype=bool)
        tiles, blk = [], 0
        h16 = AF.h16_copies()  # mixed mode: f16 forward copies, rewritten in the same tile pass
        h16_keys = []
        for (w, dst, dstT, R, C, ldT, limT) in groups:
            i = index.get(w.data_ptr())
            if i is None or covered[i] or self.params[i].numel() != R * C:
                continue  # not ours (frozen) -- or an alias we cannot express: the ordinary re-cast must stay
            covered[i] = True
            tiles_c = (C + 63) // 64
            d16 = 0
            if w.data_ptr() in h16 and h16[w.data_ptr()][0][1] == (R, C):
                h16_keys.append(h16[w.data_ptr()][0])
                d16 = h16[w.data_ptr()][1].data_ptr()
            tiles.append((i, dst.data_ptr() if dst is not None else 0, dstT.data_ptr() if dstT is not None else 0,
                          R, C, ldT, blk, tiles_c, limT, d16))
            blk += ((max(R, limT) + 63) // 64) * tiles_c
        ours = sum(1 for g in groups if g[0].data_ptr() in index)
        if not tiles or len(tiles) != ours:
            self._cast_plan = (gen, None)
            return self._cast_plan
        trow = np.zeros(len(tiles), dtype=np.dtype([("ptr", "<u8", 6), ("i", "<i4", 8)]))
        for k, (i, d, dT, R, C, ldT, b0, tc, limT, d16) in enumerate(tiles):
            trow["ptr"][k] = (self._rows[i, 0], 0, self._rows[i, 2], self._rows[i, 3], d, dT)
            trow["i"][k] = (R, C, ldT, b0, tc, limT, 0, 0)
            trow["i"][k].view(np.uint32)[6:8] = (d16 & 0xFFFFFFFF, d16 >> 32)  # the two spare words hold the f16 copy's address
        lin = self._rows[~covered].copy()
        nb = (lin[:, 4].astype(np.int64) + _CHUNK - 1) // _CHUNK
        lin[:, 5] = (np.cumsum(nb) - nb).astype(np.uint64)
        self._cast_plan = (gen, covered, lin, int(nb.sum()), trow, blk, np.array([t[0] for t in tiles]), h16_keys)
        return self._cast_plan

    def _table(self, grads, plan=None):
        key = tuple([g.data_ptr() for g in grads])
        if plan is not None:
            key = key + plan[0]  # cache generation: (invalidations, registered copies)
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if capturing:
            # a captured step never shares a table with an eager step, even at equal gradient addresses (data-parallel bucket
            # views are static): its device table and scratch must come from the graph's own memory pool
            key = key + ("captured",)
        ent = self._tables.get(key)
        if ent is None:
            assert all(g.dtype == torch.float32 and g.is_contiguous() and g.numel() == p.numel()
                       for g, p in zip(grads, self.params)), "gradients must be dense f32 of the parameter's size"
            rows = self._rows.copy()
            rows[:, 1] = key[:len(grads)]
            blk = self._blocks
            blob = rows.reshape(-1).view(np.uint8)
            if plan is not None:
                _, covered, lin, lin_blk, trow, tile_blk, tidx, _ = plan
                lin = lin.copy()
                lin[:, 1] = rows[~covered, 1]
                trow = trow.copy()
                trow["ptr"][:, 1] = rows[tidx, 1]
                blob = np.concatenate([blob, lin.reshape(-1).view(np.uint8), trow.view(np.uint8).reshape(-1)])
            if not capturing and len(self._tables) >= 16:  # eager address churn: recycle the oldest eager tables
                for k in [k for k, e in self._tables.items() if not e[5]][:8]:
                    self._free_host.append(self._tables.pop(k)[0])
            if not self._free_host:
                if capturing:
                    raise RuntimeError("FusedAdamW: out of pre-pinned table buffers under hipGraph capture")
                self._free_host.append(self._new_host())
            host = self._free_host.pop()
            host.numpy()[:blob.size] = blob  # plain host memcpy into the pinned buffer
            dev = torch.empty(host.numel(), dtype=torch.uint8, device=self.device)
            from .graph_step import capture_token

            ent = self._tables[key] = (host, dev, len(self.params), blk,
                                       torch.empty(blk, dtype=torch.float32, device=self.device), capturing,
                                       capture_token() if capturing else None)
            # (re)sent on every use below: under hipGraph capture the copy becomes a node reading THIS pinned buffer
        host, dev, n, blk, partial = ent[:5]
        dev.copy_(host, non_blocking=True)
        return dev, n, blk, partial

    def release_captured(self, token):
        """A hipGraph captured under graph_step.capture_token() == token is gone (evicted, or its capture failed): its pinned
        pointer table goes back to the free list (StepGraphs on_evict).  The caller guarantees the graph no longer runs."""
        for k in [k for k, e in self._tables.items() if e[5] and e[6] == token and token is not None]:
            self._free_host.append(self._tables.pop(k)[0])

    def captured_tables(self):
        return sum(1 for e in self._tables.values() if e[5])

    @torch.no_grad()
    def step(self):
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("FusedAdamW.step(): every trainable parameter needs a gradient")
        from . import functional as AF

        plan = self._plan() if self.cast_weights else None
        if plan is not None and plan[1] is None:
            plan = None
        table, n, blk, partial = self._table(grads, plan)
        if plan is None:
            ops.call("avsr_adamw_step", ops._ptr(table), n, blk, ops._ptr(partial), ops._ptr(self.state), self.lr,
                     self.betas[0], self.betas[1], self.eps, self.weight_decay, self.max_grad_norm, self.warmup_steps,
                     self.total_steps, ops._stream(table), nbytes=self._algo_bytes(False))
            self._note_stale(AF, False)  # parameters changed behind `_version`: their cached bf16 copies are stale
            return

        gen, _, lin, lin_blk, trow, tile_blk, _, h16_keys = plan
        base = table.data_ptr()
        lin_ptr, tile_ptr = base + 48 * n, base + 48 * n + 48 * len(lin)
        ops.call("avsr_adamw_cast_step", base, n, blk, lin_ptr, len(lin), lin_blk, tile_ptr, len(trow), tile_blk,
                 ops._ptr(partial), ops._ptr(self.state), self.lr, self.betas[0], self.betas[1], self.eps,
                 self.weight_decay, self.max_grad_norm, self.warmup_steps, self.total_steps, ops._stream(table),
                 nbytes=self._algo_bytes(True))
        AF.claim_weight_casts(self, gen)  # refresh_weight_cache() now has nothing to do for the Linear copies
        AF.claim_h16_copies(h16_keys)     # ... nor for the f16 forward copies this pass rewrote (mixed mode)
        self._note_stale(AF, True)        # ... but the conv-weight permutes are stale until it runs

    def _algo_bytes(self, cast):
        """Algorithmic HBM bytes of one step: the norm pass reads every gradient (4 B), the update reads p, g, m, v and
        writes p, m, v (28 B); with cast_weights each Linear weight also gets its two bf16 copies written (4 B)."""
        n = float(sum(p.numel() for p in self.params))
        return 32.0 * n + (4.0 * n if cast else 0.0)

    def _note_stale(self, AF, linear_rewritten):
        """Tell the weight cache which of its copies this step left behind (only copies of OUR parameters count)."""
        gen, lin, conv = AF.cached_weight_ptrs()
        if getattr(self, "_stale_gen", None) != gen:
            mine = {p.data_ptr() for p in self.params}
            self._stale_gen, self._stale_kind = gen, (bool(mine & lin), bool(mine & conv))
        has_lin, has_conv = self._stale_kind
        if has_lin and not linear_rewritten:
            AF.note_optimizer_step(False)
        elif has_conv or has_lin:
            AF.note_optimizer_step(True)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    # -- host-side views of the device state (synchronise; for logging / checkpoints only)
    @property
    def step_count(self):
        return int(self.state[0].item())

    @property
    def last_lr(self):
        return float(self.state[1].item())

    @property
    def last_grad_norm(self):
        return float(self.state[2].item())

    def state_dict(self):
        return {"state": self.state.cpu(), "exp_avg": [t.cpu() for t in self.exp_avg],
                "exp_avg_sq": [t.cpu() for t in self.exp_avg_sq],
                "hyper": dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                              max_grad_norm=self.max_grad_norm, warmup_steps=self.warmup_steps,
                              total_steps=self.total_steps)}

    def load_state_dict(self, sd):
        self.state.copy_(sd["state"])
        for dst, src in zip(self.exp_avg, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            dst.copy_(src)


class ShardedAdamW:
    """The same optimizer step (global-norm clip + AdamW + per-step warm-up cosine: lightning.py:48-52, train.py:41, cosine.py)
    SHARDED over the data-parallel ranks -- SURVEY section 5 / 8e, the round-5 verdict's item 7.  `buckets` is a
    `ddp.GradBuckets(shard=True)`: parameters and gradients live in flat per-bucket buffers, the backward pass ends with a
    reduce-scatter, and this rank then holds the averaged gradients of slice `rank` of every bucket.  One step:

        sum of squares of this rank's gradient slices (avsr_multi_sumsq)  ->  all-reduce of ONE float  ->  clip coefficient, step
        counter, learning rate + AdamW on this rank's parameter slices (avsr_adamw_apply; moments exist for the slices only)  ->
        all-gather of every flat parameter buffer (in place: each rank contributes the slice it just updated).

    Per rank the optimizer moves 1 / N of its 36 B per parameter (1.5 ms -> ~0.2 ms at N = 8 on the 250 M-parameter model); the
    wire carries what a ring all-reduce carries (reduce-scatter + all-gather ARE its two halves), but the second half now sits
    AFTER the optimizer, where only the next step's forward pass can hide it -- this version does not overlap it (the compute
    stream waits for the gather), so the mode is opt-in (AVSR_SHARD_OPT=1 with AVSR_DDP=buckets) until a multi-GPU run prices
    it; DESIGN.md section 6 holds the budget.  Every element is updated by exactly ONE rank and copied to the others: replicas are
    bit-identical by construction, and equal to the unsharded FusedAdamW step element for element (the update is element-wise;
    only the summation order of the gradient norm differs, which matters when clipping is active: ~1e-7 relative)."""

    collective_state = True  # state_dict() is a collective: every rank must call it

    def __init__(self, buckets, lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0, warmup_steps=0, total_steps=0):
        assert getattr(buckets, "shard", False), "ShardedAdamW needs ddp.GradBuckets(shard=True)"
        self.buckets = buckets
        self.device = buckets.device
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.max_grad_norm, self.warmup_steps, self.total_steps = float(max_grad_norm), int(warmup_steps), int(total_steps)
        if self.total_steps > 0 and self.total_steps <= self.warmup_steps:
            raise ValueError(f"ShardedAdamW: total_steps ({self.total_steps}) must exceed warmup_steps ({self.warmup_steps})")
        self.state = torch.zeros(4, dtype=torch.float32, device=self.device)  # step, lr, grad norm, clip coefficient
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        rows, blk = [], 0
        self.exp_avg, self.exp_avg_sq = [], []
        for b in range(len(buckets.flat)):
            lo, hi = buckets.shard_range(b)
            m = torch.zeros(hi - lo, dtype=torch.float32, device=self.device)
            v = torch.zeros(hi - lo, dtype=torch.float32, device=self.device)
            self.exp_avg.append(m)
            self.exp_avg_sq.append(v)
            rows.append((buckets.pflat[b][lo:hi].data_ptr(), buckets.flat[b][lo:hi].data_ptr(), m.data_ptr(), v.data_ptr(), hi - lo, blk))
            blk += (hi - lo + _CHUNK - 1) // _CHUNK
        self._blocks = blk
        tab = np.array(rows, dtype=np.uint64)  # (blk0 in the low half of the last u64, pad 0: csrc/optim.hip OptEntry)
        self._table = torch.from_numpy(tab.reshape(-1).view(np.uint8).copy()).to(self.device)
        self._n = len(rows)
        self._partial = torch.empty(max(blk, 1), dtype=torch.float32, device=self.device)
        # (this object keeps no per-step host tables: the addresses above are those of persistent buffers -- nothing to pin per
        # captured graph, nothing to release when one is evicted)

    def release_captured(self, token):
        pass

    def captured_tables(self):
        return 0

    @torch.no_grad()
    def step(self):
        from . import functional as AF

        bk = self.buckets
        st = ops._stream(self._table)
        ops.call("avsr_multi_sumsq", ops._ptr(self._table), self._n, self._blocks, ops._ptr(self._partial), ops._ptr(self.sumsq), st,
                 nbytes=4.0 * sum(m.numel() for m in self.exp_avg))
        if bk.world > 1:
            if bk.comm is not None:
                bk.comm.all_reduce(self.sumsq)
            else:
                import torch.distributed as dist

                dist.all_reduce(self.sumsq, group=bk.group)
        ops.call("avsr_adamw_apply", ops._ptr(self._table), self._n, self._blocks, ops._ptr(self.sumsq), ops._ptr(self.state), self.lr,
                 self.betas[0], self.betas[1], self.eps, self.weight_decay, self.max_grad_norm, self.warmup_steps, self.total_steps, st,
                 nbytes=28.0 * sum(m.numel() for m in self.exp_avg))
        if bk.world > 1:
            for b, pf in enumerate(bk.pflat):  # every rank's freshly updated slice to everybody (in place)
                lo, hi = bk.shard_range(b)
                if bk.comm is not None:
                    bk.comm.all_gather(pf, pf[lo:hi])
                else:
                    import torch.distributed as dist

                    dist.all_gather_into_tensor(pf, pf[lo:hi].clone(), group=bk.group)
        # the parameters changed behind their tensors' version counters: every cached operand copy is stale
        AF.note_optimizer_step(False)

    def zero_grad(self, set_to_none=True):
        for p in self.buckets.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @property
    def step_count(self):
        return int(self.state[0].item())

    @property
    def last_lr(self):
        return float(self.state[1].item())

    @property
    def last_grad_norm(self):
        return float(self.state[2].item())

    def _full(self, shards):
        """Collective: the per-parameter tensors (FusedAdamW's layout) of a moment held as one slice per bucket and rank."""
        bk = self.buckets
        out = [None] * len(bk.params)
        for b, sh in enumerate(shards):
            full = torch.empty_like(bk.flat[b])
            if bk.world > 1:
                if bk.comm is not None:
                    bk.comm.all_gather(full, sh)
                else:
                    import torch.distributed as dist

                    dist.all_gather_into_tensor(full, sh, group=bk.group)
            else:
                full.copy_(sh)
            for i in bk.members[b]:
                o, n = bk.offset[i], bk.params[i].numel()
                out[i] = full[o:o + n].view_as(bk.params[i]).cpu()
        return out

    def state_dict(self):
        """FusedAdamW's checkpoint layout (interchangeable `last.ckpt` files); a COLLECTIVE -- every rank calls it."""
        return {"state": self.state.cpu(), "exp_avg": self._full(self.exp_avg), "exp_avg_sq": self._full(self.exp_avg_sq),
                "hyper": dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                              max_grad_norm=self.max_grad_norm, warmup_steps=self.warmup_steps, total_steps=self.total_steps)}

    def load_state_dict(self, sd):
        bk = self.buckets
        self.state.copy_(sd["state"])
        for name, dst in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            for b in range(len(bk.flat)):
                lo, hi = bk.shard_range(b)
                full = torch.zeros(bk.flat[b].numel(), dtype=torch.float32)
                for i in bk.members[b]:
                    o, n = bk.offset[i], bk.params[i].numel()
                    full[o:o + n] = sd[name][i].reshape(-1)
                dst[b].copy_(full[lo:hi])
