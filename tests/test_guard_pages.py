"""Out-of-bounds detector for the kernels: every operand is placed so that it ENDS exactly at an unmapped page and BEGINS
right after one (mmap + mprotect), then the emulator build of the kernel runs on it.  An access past either end of an
operand -- which the GPU reports as a "memory access fault" and the plain emulator silently tolerates -- is a SIGSEGV of the
test process here (run in a child process, so the suite reports a failure instead of dying).

Covers the kernels written in round 2 whose footprint depends on ragged tile edges: the transposed-formulation attention
forward / backward-dq, both key / value-side backward paths (generic and the opt-in k-major tile kernel, whose first MI355X run
ended in exactly such a fault), the single-launch BatchNorm and the attention-branch target preparation."""
import multiprocessing as mp
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _child(emu_path, case, q):
    import ctypes
    import mmap

    import torch

    sys.path.insert(0, os.path.dirname(HERE))
    from auto_avsr_amd import _lib, ops

    _lib._install_for_tests(emu_path)
    libc = ctypes.CDLL(None, use_errno=True)
    PAGE = 4096
    keep = []

    def guarded(t):
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        body = (nbytes + PAGE - 1) // PAGE * PAGE
        m = mmap.mmap(-1, body + 2 * PAGE)
        addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
        for off in (0, PAGE + body):
            assert libc.mprotect(ctypes.c_void_p(addr + off), PAGE, 0) == 0
        start = PAGE + body - nbytes
        start -= start % 16  # the kernels want 16-byte aligned bases (at most 14 bytes before the guard stay unprotected)
        buf = (ctypes.c_char * nbytes).from_address(addr + start)
        g = torch.frombuffer(buf, dtype=t.dtype, count=t.numel()).view(t.shape)
        g.copy_(t)
        keep.append((m, buf))
        return g

    torch.manual_seed(3)
    kind = case[0]
    if kind == "attn":
        _, relpos, B, T, Tk, H, mkind, kv_knob = case
        D = 64
        bf = torch.bfloat16
        qu, qv = guarded(torch.randn(B, T, H, D).to(bf)), guarded(torch.randn(B, T, H, D).to(bf))
        k, v = guarded(torch.randn(B, Tk, H, D).to(bf)), guarded(torch.randn(B, Tk, H, D).to(bf))
        pos = guarded(torch.randn(2 * T - 1, H * D).to(bf)) if relpos else None
        mask = None
        if mkind == "pad":
            m = torch.ones(B, 1, Tk, dtype=torch.bool)
            m[-1, 0, Tk - 3:] = False
            mask = guarded(m)
        elif mkind == "causal":
            mask = guarded(torch.tril(torch.ones(T, Tk, dtype=torch.bool)).unsqueeze(0).expand(B, T, Tk).contiguous())
        dout = guarded(torch.randn(B, T, H * D).to(bf))
        out, lse = ops.attention_fwd(qu, qv if relpos else None, k, v, pos, mask, 0.125, drop_p=0.1, seed=4)
        out, lse = guarded(out), guarded(lse)
        ops.tune(10, kv_knob)
        try:
            dk, dv = guarded(torch.zeros(B, Tk, H, D, dtype=bf)), guarded(torch.zeros(B, Tk, H, D, dtype=bf))
            dpos = guarded(torch.zeros(2 * T - 1, H * D)) if relpos else None
            kw = dict(dk_out=dk, dv_out=dv)
            if relpos:
                dq = guarded(torch.zeros(B, T, H, D, dtype=bf))
                du, dvb = guarded(torch.zeros(H * D)), guarded(torch.zeros(H * D))
                kw.update(dpos_out=dpos, dq_sum=dq, du=du, dv_bias=dvb)
            ops.attention_bwd(qu, qv if relpos else None, k, v, pos, mask, out, lse, dout, 0.125, drop_p=0.1, seed=4, **kw)
        finally:
            ops.tune(10, 0)
        q.put(float(dk.float().abs().sum()))
    elif kind == "bn_small":
        _, rows, C, dtype = case
        x = guarded((torch.randn(rows, C) * 2).to(dtype))
        dy = guarded(torch.randn(rows, C).to(dtype))
        g, b = guarded(torch.rand(C) + 0.5), guarded(torch.randn(C))
        rm, rv = guarded(torch.zeros(C)), guarded(torch.ones(C))
        nbt = guarded(torch.zeros(1, dtype=torch.int64))
        y, mean, invstd = ops.bn_small_fwd(x, rows, C, g, b, 1e-5, 0.1, rm, rv, nbt, 1)
        dx, dg, db = ops.bn_small_bwd(x, dy, rows, C, guarded(mean), guarded(invstd), g, b, 1)
        q.put(float(dx.float().abs().sum()))
    elif kind == "targets":
        _, B, L = case
        ys = torch.randint(1, 50, (B, L))
        ys[0, L // 2:] = -1
        r = ops.prepare_targets(guarded(ys), 60, 60, -1)
        q.put(float(r[3].sum()))
    else:
        raise AssertionError(kind)


CASES = [
    ("attn", True, 2, 100, 100, 2, "pad", 0),
    ("attn", True, 2, 100, 100, 2, "pad", 2),     # opt-in k-major key / value kernel
    ("attn", True, 1, 129, 129, 1, None, 2),
    ("attn", False, 2, 33, 130, 2, "pad", 0),
    ("attn", False, 2, 33, 130, 2, "pad", 2),
    ("attn", False, 2, 65, 65, 2, "causal", 0),
    ("attn", False, 2, 65, 65, 2, "causal", 2),
    ("bn_small", 1, 8, "f32"), ("bn_small", 513, 40, "bf16"), ("bn_small", 2048, 16, "f32"),
    ("targets", 3, 7), ("targets", 2, 300),
]


@pytest.mark.parametrize("case", CASES, ids=[("-".join(str(c) for c in cs)) for cs in CASES])
def test_no_access_outside_the_operands(emu_lib_path, case):
    import torch

    case = tuple({"f32": torch.float32, "bf16": torch.bfloat16}.get(c, c) if isinstance(c, str) else c for c in case)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_child, args=(emu_lib_path, case, q))
    p.start()
    p.join(300)
    assert not p.is_alive(), "kernel did not finish"
    assert p.exitcode == 0, f"child died with exit code {p.exitcode} (SIGSEGV = -11: an access outside an operand)"
    val = q.get(timeout=10)
    assert val == val  # finite / not NaN
