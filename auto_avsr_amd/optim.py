"""Fused optimizer step for the native training loop: global-norm gradient clipping, AdamW and the per-step warm-up
cosine learning-rate schedule of the reference (lightning.py:48-52: AdamW(lr, betas=(0.9, 0.98), weight_decay);
train.py:41: gradient_clip_val=10.0; cosine.py:6-25) as THREE kernel launches over all parameters
(csrc/optim.hip: avsr_adamw_step), with the step count, learning rate, gradient norm and clip coefficient resident on the
device -- no host synchronisation, capturable in a hipGraph."""
import numpy as np
import torch

from . import ops

_CHUNK = 4096


class FusedAdamW:
    def __init__(self, params, lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0, warmup_steps=0,
                 total_steps=0):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        assert all(p.dtype == torch.float32 and p.is_contiguous() for p in self.params), "f32 contiguous master weights"
        self.device = self.params[0].device
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.max_grad_norm, self.warmup_steps, self.total_steps = float(max_grad_norm), int(warmup_steps), int(total_steps)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.state = torch.zeros(4, dtype=torch.float32, device=self.device)  # step, lr, grad norm, clip coefficient
        self._tables = {}  # gradient addresses -> (pinned host table, device table, n, blocks, scratch, made under capture)
        # pinned staging buffers are allocated HERE: hipHostMalloc is not allowed while a stream is capturing, and the
        # pointer table of a captured step can only be built during the capture (that is when its gradients exist)
        self._table_bytes = 48 * len(self.params)
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
        self._free_host = [self._new_host() for _ in range(8)]

    def _new_host(self):
        t = torch.empty(self._table_bytes, dtype=torch.uint8)
        return t.pin_memory() if self.device.type == "cuda" else t

    # -- gradient pointer table: gradients are fresh tensors after every backward, so their addresses may move
    def _table(self, grads):
        key = tuple([g.data_ptr() for g in grads])
        ent = self._tables.get(key)
        if ent is None:
            assert all(g.dtype == torch.float32 and g.is_contiguous() and g.numel() == p.numel()
                       for g, p in zip(grads, self.params)), "gradients must be dense f32 of the parameter's size"
            rows = self._rows.copy()
            rows[:, 1] = key
            blk = self._blocks
            capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
            if not capturing and len(self._tables) >= 16:  # eager address churn: recycle the oldest eager tables
                for k in [k for k, e in self._tables.items() if not e[5]][:8]:
                    self._free_host.append(self._tables.pop(k)[0])
            if not self._free_host:
                if capturing:
                    raise RuntimeError("FusedAdamW: out of pre-pinned table buffers under hipGraph capture")
                self._free_host.append(self._new_host())
            host = self._free_host.pop()
            host.numpy()[:] = rows.reshape(-1).view(np.uint8)  # plain host memcpy into the pinned buffer
            dev = torch.empty(host.numel(), dtype=torch.uint8, device=self.device)
            ent = self._tables[key] = (host, dev, len(self.params), blk,
                                       torch.empty(blk, dtype=torch.float32, device=self.device), capturing)
            # (re)sent on every use below: under hipGraph capture the copy becomes a node reading THIS pinned buffer
        host, dev, n, blk, partial, _ = ent
        dev.copy_(host, non_blocking=True)
        return dev, n, blk, partial

    @torch.no_grad()
    def step(self):
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            raise RuntimeError("FusedAdamW.step(): every trainable parameter needs a gradient")
        table, n, blk, partial = self._table(grads)
        ops.call("avsr_adamw_step", ops._ptr(table), n, blk, ops._ptr(partial), ops._ptr(self.state), self.lr,
                 self.betas[0], self.betas[1], self.eps, self.weight_decay, self.max_grad_norm, self.warmup_steps,
                 self.total_steps, ops._stream(table))

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
