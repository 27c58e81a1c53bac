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
* `finish()` makes the compute stream wait for the outstanding reductions (before the optimizer reads the gradients).

Why not torch DDP: its reducer cannot be captured into a hipGraph on this stack (`tools/rccl_capture_probe.py`: plain RCCL
all-reduce / all-gather capture and replay fine, `DistributedDataParallel`'s backward invalidates the capture), which left
the N > 1 step on eager launches -- host-limited.  Everything this class issues is a stream operation:
the whole data-parallel step, collectives included, replays as one graph per batch shape.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import ops

_CHUNK = 4096  # elements per block of avsr_multi_copy_scale (csrc/optim.hip OPT_CHUNK)


class GradBuckets:
    def __init__(self, params, group=None, bucket_mb=64.0):
        self.params = [p for p in params if p.requires_grad]
        assert self.params and all(p.dtype == torch.float32 for p in self.params)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = self.params[0].device
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        # buckets over the parameters in reverse order; every slice starts 16-byte aligned
        self.bucket_of, self.offset, sizes, members = {}, {}, [], []
        cur, cur_n = [], 0
        for i in reversed(range(len(self.params))):
            n = (self.params[i].numel() + 3) // 4 * 4
            if cur and cur_n + n > cap:
                members.append(cur)
                sizes.append(cur_n)
                cur, cur_n = [], 0
            self.bucket_of[i], self.offset[i] = len(members), cur_n
            cur.append(i)
            cur_n += n
        members.append(cur)
        sizes.append(cur_n)
        self.members = members
        self.flat = [torch.zeros(n, dtype=torch.float32, device=self.device) for n in sizes]
        self.views = {i: self.flat[self.bucket_of[i]][self.offset[i]: self.offset[i] + self.params[i].numel()].view_as(self.params[i])
                      for i in range(len(self.params))}
        self._left = [len(m) for m in members]
        self._works = []
        self._tables = {}  # (bucket, gradient addresses) -> (pinned host rows, device table, blocks)
        # pinned staging for the pointer tables is allocated HERE (hipHostMalloc is not allowed under stream capture)
        self._free_host = [[self._new_host(len(m)) for _ in range(24)] for m in members]
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]

    def _new_host(self, n):
        t = torch.empty(48 * n, dtype=torch.uint8)
        return t.pin_memory() if self.device.type == "cuda" else t

    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            self._left[b] -= 1
            if self._left[b] == 0:
                self._flush(b)
        return hook

    def _flush(self, b):
        idx = self.members[b]
        grads = [self.params[i].grad for i in idx]
        assert all(g is not None and g.dtype == torch.float32 and g.is_contiguous() for g in grads)
        key = (b,) + tuple(g.data_ptr() for g in grads)
        ent = self._tables.get(key)
        if ent is None:
            rows = np.zeros((len(idx), 6), dtype=np.uint64)
            numel = np.array([self.params[i].numel() for i in idx], dtype=np.int64)
            blocks = (numel + _CHUNK - 1) // _CHUNK
            rows[:, 0] = [self.views[i].data_ptr() for i in idx]
            rows[:, 1] = key[1:]
            rows[:, 4] = numel.astype(np.uint64)
            rows[:, 5] = (np.cumsum(blocks) - blocks).astype(np.uint64)
            capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
            if not capturing and len(self._tables) >= 64:  # eager address churn: recycle the eager tables
                for k in [k for k, e in self._tables.items() if not e[3]]:
                    self._free_host[k[0]].append(self._tables.pop(k)[0])
            if not self._free_host[b]:
                if capturing:
                    raise RuntimeError("GradBuckets: out of pre-pinned table buffers under hipGraph capture")
                self._free_host[b].append(self._new_host(len(idx)))
            host = self._free_host[b].pop()
            host.numpy()[:] = rows.reshape(-1).view(np.uint8)
            ent = self._tables[key] = (host, torch.empty(host.numel(), dtype=torch.uint8, device=self.device), int(blocks.sum()),
                                       capturing)
        host, dev, blocks, _ = ent
        dev.copy_(host, non_blocking=True)  # (under capture: a memcpy node reading this pinned buffer on every replay)
        ops.call("avsr_multi_copy_scale", ops._ptr(dev), len(idx), blocks, 1.0 / self.world, ops._stream(dev),
                 nbytes=8.0 * self.flat[b].numel())
        for i in idx:
            self.params[i].grad = self.views[i]
        if self.world > 1 or self.group is not None:
            self._works.append(dist.all_reduce(self.flat[b], group=self.group, async_op=True))

    def finish(self):
        """After loss.backward(): every bucket has been flushed; the compute stream waits for the reductions."""
        missing = [b for b, n in enumerate(self._left) if n != 0]
        if missing:
            raise RuntimeError(f"GradBuckets.finish(): buckets {missing} did not receive all of their gradients "
                               "(a parameter without gradient -- find_unused_parameters=False semantics)")
        for w in self._works:
            w.wait()
        self._works.clear()
        self._left = [len(m) for m in self.members]

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
