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
    elif kind == "gemm_nt":
        _, M, N, K, tile = case
        bf = torch.bfloat16
        A, Bw = guarded(torch.randn(M, K).to(bf)), guarded(torch.randn(N, K).to(bf))
        bias = guarded(torch.randn(N))
        C = guarded(torch.zeros(M, N, dtype=bf))
        ops.gemm_bf16_nt(A, K, Bw, K, M, N, K, C, N, bias=bias, act=1, tile=tile)
        q.put(float(C.float().abs().sum()))
    elif kind == "gemm_tn":
        _, M, N, K, split = case
        bf = torch.bfloat16
        A, Bm = guarded(torch.randn(K, M).to(bf)), guarded(torch.randn(K, N).to(bf))
        C = guarded(torch.zeros(M, N))
        cs = guarded(torch.zeros(M))
        ops.gemm_bf16_tn(A, M, Bm, N, M, N, K, C, N, accumulate=split > 1, split_k=split, colsum_a=cs)
        q.put(float(C.abs().sum()))
    elif kind == "layernorm":
        _, rows, cols = case
        x, dy = guarded(torch.randn(rows, cols)), guarded(torch.randn(rows, cols).bfloat16())
        g, b = guarded(torch.rand(cols) + 0.5), guarded(torch.randn(cols))
        y, mean, rstd = ops.layernorm_fwd(x, g, b, torch.bfloat16)
        dg, db = guarded(torch.zeros(cols)), guarded(torch.zeros(cols))
        gout, gsum = guarded(torch.zeros(rows, cols, dtype=torch.bfloat16)), guarded(torch.zeros(cols))
        dx = ops.layernorm_bwd(dy, x, g, guarded(mean), guarded(rstd), dg, db, gout=gout, gsum=gsum, alpha=0.5, drop_p=0.1, seed=3)
        q.put(float(dx.abs().sum()))
    elif kind == "dwconv":
        _, B, T, C, K = case
        bf = torch.bfloat16
        a = guarded(torch.randn(B * T, 2 * C).to(bf))
        w, bias = guarded(torch.randn(C, K)), guarded(torch.randn(C))
        c = ops.dwconv(a, w, bias, B, T, C, K, glu_in=True)
        dc = guarded(torch.randn(B * T, C).to(bf))
        dw, dbias = guarded(torch.zeros(C, K)), guarded(torch.zeros(C))
        ops.dwconv_wgrad(a, dc, dw, dbias, B, T, C, K, glu_in=True)
        da = ops.dwconv(dc, w, None, B, T, C, K, flip=True, glu_a=a)
        q.put(float(da.float().abs().sum()) + float(c.float().abs().sum()))
    elif kind == "ctc":
        _, B, T, V, L = case
        ld = (V + 7) // 8 * 8
        lg = torch.zeros(B * T, ld)
        lg[:, :V] = torch.randn(B * T, V)
        labels = torch.randint(1, V, (B, L))
        labels[-1, L // 2:] = -1
        in_lens = torch.full((B,), T, dtype=torch.int64)
        in_lens[-1] = max(T - 5, 1)
        nll, grad = ops.ctc_loss(guarded(lg), ld, guarded(labels), guarded(in_lens), B, T, V)
        q.put(float(grad.abs().sum()))
    elif kind == "targets":
        _, B, L = case
        ys = torch.randint(1, 50, (B, L))
        ys[0, L // 2:] = -1
        r = ops.prepare_targets(guarded(ys), 60, 60, -1)
        q.put(float(r[3].sum()))
    elif kind == "conv_patch":  # round 6, csrc/conv_patch.hip: whole-image tiles, ragged last tile, zero slot, offset tables
        _, N, H, Cin, Cout = case
        bf = torch.bfloat16
        w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
        x = guarded(torch.randn(N, H, H, Cin).to(bf))
        dy = guarded(torch.randn(N, H, H, Cout).to(bf))
        res = guarded(torch.randn(N, H, H, Cin).to(bf))
        wp = guarded(ops.conv_weight_permute(w, bf))
        wpd = guarded(ops.conv_weight_permute(w, bf, to_dgrad=True))
        xh = guarded(torch.randn(N, H, H, Cin).half())
        w16 = guarded((0.05 * torch.randn(Cout, 2, 9 * Cin)).half())
        tot = 0.0
        try:
            for knob in (2, 3):  # the patch-staged kernel whatever the grid size, two / three weight stages
                ops.tune(20, knob)
                y = ops.conv2d_fwd(x, wp, N, H, H, Cin, Cout, 3, 3, 1, 1, 1, False)
                y2 = ops.conv2d_fwd(xh, w16, N, H, H, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=2)
                y1 = ops.conv2d_fwd(xh, w16, N, H, H, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=1)
                tot += float(y.float().abs().sum()) + float(y2.float().abs().sum()) + float(y1.float().abs().sum())
                if Cin % 128 == 0:
                    dx = ops.conv2d_dgrad(dy, wpd, res, N, H, H, Cin, Cout, 3, 3, 1, 1, 1, False)
                    tot += float(dx.float().abs().sum())
        finally:
            ops.tune(20, 0)
        q.put(tot)
    elif kind == "convmod_fused":  # round 6, csrc/convmod_fused.hip (opt-in)
        _, B, T, C, K, dtype = case
        rows = B * T
        a = guarded((torch.randn(rows, 2 * C) * 1.2).to(dtype))
        w, bias = guarded(torch.randn(C, K) * 0.3), guarded(torch.randn(C) * 0.1)
        g, b = guarded(torch.rand(C) + 0.5), guarded(torch.randn(C) * 0.1)
        rm, rv = guarded(torch.zeros(C)), guarded(torch.ones(C))
        nbt = guarded(torch.zeros(1, dtype=torch.int64))
        s_, c, mean, invstd = ops.convmod_dwbn_fwd(a, w, bias, B, T, C, K, g, b, 1e-5, 0.1, rm, rv, nbt)
        ds = guarded(torch.randn(rows, C).to(dtype))
        dw, dbias = guarded(torch.zeros(C, K)), guarded(torch.zeros(C))
        da, dg, db = ops.convmod_dwbn_bwd(a, guarded(c), ds, guarded(mean), guarded(invstd), g, b, w, B, T, C, K, dw, dbias)
        q.put(float(da.float().abs().sum()) + float(s_.float().abs().sum()))
    elif kind == "attn_ksplit":  # round 6: key range split over two wave groups (forced: knob 8 = 3), source-attention geometry
        _, B, T, Tk, H = case
        D = 64
        bf = torch.bfloat16
        qu = guarded(torch.randn(B, T, H, D).to(bf))
        k, v = guarded(torch.randn(B, Tk, H, D).to(bf)), guarded(torch.randn(B, Tk, H, D).to(bf))
        m = torch.ones(B, 1, Tk, dtype=torch.bool)
        m[-1, 0, Tk - 37:] = False
        try:
            ops.tune(8, 3)
            out, lse = ops.attention_fwd(qu, None, k, v, None, guarded(m), 0.125)
        finally:
            ops.tune(8, 0)
        q.put(float(out.float().abs().sum()))
    elif kind == "permute_split8":  # round 6: split8 conv-weight copies from the table launch
        import struct

        _, Cout, Cin, taps = case
        w = guarded(torch.randn(Cout, Cin, taps))
        o = guarded(torch.zeros(Cout, taps * Cin))
        blob = struct.pack("<QQiiiiiiii", w.data_ptr(), o.data_ptr(), Cout, Cin, taps, 0, 0, 3, 0, 0)
        table = guarded(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        ops.multi_weight_permute(table, 1, ops.weight_permute_blocks(Cout, Cin, False), taps)
        q.put(float(o.abs().sum()))
    else:
        raise AssertionError(kind)


CASES = [
    ("attn", True, 2, 100, 100, 2, "pad", 1),     # generic key / value path
    ("attn", True, 2, 100, 100, 2, "pad", 0),     # k-major key / value kernel (default)
    ("attn", True, 1, 129, 129, 1, None, 0),
    ("attn", False, 2, 33, 130, 2, "pad", 1),
    ("attn", False, 2, 33, 130, 2, "pad", 0),
    ("attn", False, 2, 65, 65, 2, "causal", 1),
    ("attn", False, 2, 65, 65, 2, "causal", 0),
    ("bn_small", 1, 8, "f32"), ("bn_small", 513, 40, "bf16"), ("bn_small", 2048, 16, "f32"),
    ("targets", 3, 7), ("targets", 2, 300),
    # long-standing kernels with ragged edges (clamped / zero-page loads, pitched rows): cheap to keep under the same guard
    ("gemm_nt", 100, 72, 128, 1), ("gemm_nt", 130, 200, 64, 7), ("gemm_nt", 257, 136, 192, 4),
    ("gemm_tn", 72, 64, 100, 1), ("gemm_tn", 128, 192, 333, 2),
    ("layernorm", 37, 256), ("layernorm", 1030, 768),
    ("dwconv", 2, 50, 64, 31), ("dwconv", 3, 129, 128, 7),
    ("ctc", 2, 40, 53, 9), ("ctc", 1, 150, 301, 70),
    # round 6
    ("conv_patch", 30, 3, 128, 128), ("conv_patch", 5, 6, 64, 128), ("conv_patch", 9, 6, 128, 256), ("conv_patch", 3, 11, 128, 128),
    ("convmod_fused", 3, 37, 16, 31, "f32"), ("convmod_fused", 4, 512, 8, 15, "bf16"), ("convmod_fused", 1, 1, 8, 31, "f32"),
    ("attn_ksplit", 2, 33, 300, 2), ("attn_ksplit", 1, 65, 257, 1),
    ("permute_split8", 72, 64, 9), ("permute_split8", 128, 128, 3),
]


def _run_all(emu_path, cases, progress_path):
    """Child process: all cases one after the other; the id of the case being run is on disk before the kernels start."""
    import queue as _q

    for cs in cases:
        with open(progress_path, "w") as f:
            f.write("-".join(str(c) for c in cs))
        _child(emu_path, cs, _q.Queue())
    with open(progress_path, "w") as f:
        f.write(f"done:{len(cases)}")


def test_no_access_outside_the_operands(emu_lib_path, tmp_path):
    """One child process runs every case (process start-up dominates a case); if it dies, the progress file names the case."""
    import torch

    cases = [tuple({"f32": torch.float32, "bf16": torch.bfloat16}.get(c, c) if isinstance(c, str) and c in ("f32", "bf16") else c
                   for c in cs) for cs in CASES]
    progress = str(tmp_path / "progress.txt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_run_all, args=(emu_lib_path, cases, progress))
    p.start()
    p.join(900)
    assert not p.is_alive(), "kernels did not finish"
    last = open(progress).read() if os.path.exists(progress) else "(nothing started)"
    assert p.exitcode == 0 and last == f"done:{len(CASES)}", \
        f"child exit code {p.exitcode} in case {last} (SIGSEGV = -11: an access outside an operand)"
