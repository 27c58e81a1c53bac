"""Stream-ordered collectives of the data-parallel step (csrc/comm.hip): RCCL's C API called from `libavsr_hip.so`, up to four
communicators per process, every operation ONE operation on the current HIP stream -- no torch Work objects, no events for a
process-group watchdog to poll, hence safe inside a hipGraph capture.

    comm = StreamComm.from_process_group()      # bootstrap: rank 0's 128-byte RCCL id travels over torch.distributed
    comm.all_reduce(flat_f32)                   # in place, sum, on torch.cuda.current_stream()
    comm.all_gather(out_f32, mine_f32)

Users: `ddp.GradBuckets(comm=...)` (gradient buckets on a side stream), `functional.set_bn_sync(group, comm=...)` (the
cross-rank BatchNorm statistics), `bench.py --ddp buckets-graph` (W / sum(B) all-gather).  GPU only: the CPU test suite runs the
same callers on torch.distributed / gloo (comm=None)."""
import ctypes

import torch

from . import _lib, ops


class StreamComm:
    _live = {}  # slot -> communicator (the library holds up to four)

    def __init__(self, unique_id: bytes, world: int, rank: int):
        assert len(unique_id) == 128
        free = [s for s in range(4) if s not in StreamComm._live]
        if not free:
            raise RuntimeError("StreamComm: four communicators per process (close() one)")
        self.slot = free[0]
        buf = ctypes.create_string_buffer(unique_id, 128)
        _lib.lib().call("avsr_comm_init", self.slot, ctypes.cast(buf, ctypes.c_void_p).value, world, rank)
        self.world, self.rank = world, rank
        StreamComm._live[self.slot] = self

    @staticmethod
    def new_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _lib.lib().call("avsr_comm_unique_id", ctypes.cast(buf, ctypes.c_void_p).value)
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None):
        """Collective over the torch process group `group` (default: WORLD): every rank must call it."""
        import torch.distributed as dist

        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], world, rank)

    @classmethod
    def single(cls):
        """A one-rank communicator (tools/rccl_capi_world1.py: the whole N > 1 code path on one GPU)."""
        return cls(cls.new_unique_id(), 1, 0)

    def all_reduce(self, t):
        assert t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous() and t.is_cuda
        ops.call("avsr_comm_all_reduce", self.slot, ops._ptr(t), t.numel(), 0 if t.dtype == torch.float32 else 1, ops._stream(t),
                 nbytes=2.0 * t.numel() * t.element_size())
        return t

    def ranks(self):
        """Size of the RCCL communicator as the library reports it (bench.py prints it: config.rccl_ranks)."""
        return int(_lib.lib().call("avsr_comm_size", self.slot))

    def all_gather(self, out, mine):
        assert out.dtype == mine.dtype == torch.float32 and out.is_contiguous() and mine.is_contiguous()
        assert out.numel() == self.world * mine.numel()
        ops.call("avsr_comm_all_gather_f32", self.slot, ops._ptr(mine), ops._ptr(out), mine.numel(), ops._stream(out))
        return out

    def reduce_scatter(self, full, mine=None):
        """mine (default: this rank's slice of `full`, in place) = sum over the ranks of their `full`'s slice `rank`."""
        n = full.numel() // self.world
        assert full.numel() == n * self.world and full.is_contiguous() and full.dtype in (torch.float32, torch.bfloat16)
        if mine is None:
            mine = full[self.rank * n:(self.rank + 1) * n]
        ops.call("avsr_comm_reduce_scatter", self.slot, ops._ptr(full), ops._ptr(mine), n, 0 if full.dtype == torch.float32 else 1,
                 ops._stream(full), nbytes=float(full.numel() * full.element_size()))
        return mine

    def close(self):
        if StreamComm._live.get(self.slot) is self:
            _lib.lib().call("avsr_comm_destroy", self.slot)
            del StreamComm._live[self.slot]


class GroupComm:
    """The StreamComm interface on a torch.distributed process group of its own (any backend).  The CPU test suite runs the
    `comm=` control flow of ddp.GradBuckets, the cross-rank BatchNorm and bench.py's data-parallel step on it over gloo (the
    emulator build has no RCCL); on GPUs it is a transport of last resort (collectives through c10d: not graph-capturable)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist = dist
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.group = dist.new_group(ranks=list(range(self.world))) if group is None else group  # (collective: every rank calls it)

    @classmethod
    def from_process_group(cls, group=None):
        return cls(group)

    def all_reduce(self, t):
        if t.dtype == torch.bfloat16 and not t.is_cuda:  # (gloo has no bf16 sum: reduce the exact f32 images, round once)
            f = t.float()
            self._dist.all_reduce(f, group=self.group)
            t.copy_(f)
            return t
        self._dist.all_reduce(t, group=self.group)
        return t

    def all_gather(self, out, mine):
        assert out.numel() == self.world * mine.numel()
        if mine.untyped_storage().data_ptr() == out.untyped_storage().data_ptr():
            mine = mine.clone()  # (in-place form -- `mine` is this rank's slice of `out`: c10d wants disjoint buffers)
        if out.is_cuda and self._dist.get_backend(self.group) == "gloo":
            # (gloo has no all-gather on device tensors: sum of one-hot placements)
            out.zero_()
            out.view(self.world, -1)[self.rank].copy_(mine.reshape(-1))
            self._dist.all_reduce(out, group=self.group)
            return out
        self._dist.all_gather_into_tensor(out, mine, group=self.group)
        return out

    def reduce_scatter(self, full, mine=None):
        """(gloo has no reduce-scatter: the all-reduce, of which this rank keeps its slice)"""
        n = full.numel() // self.world
        self.all_reduce(full)
        sl = full[self.rank * n:(self.rank + 1) * n]
        if mine is not None and mine.data_ptr() != sl.data_ptr():
            mine.copy_(sl)
            return mine
        return sl

    def ranks(self):
        return self.world

    def close(self):
        pass
