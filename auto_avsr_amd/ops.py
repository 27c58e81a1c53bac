"""Thin Python wrappers over the C ABI (include/avsr_hip.h): one function per entry point,
taking torch tensors as device-memory handles.  No compute happens in Python or ATen here."""
import contextlib
import os

import torch

from . import _lib

F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}  # f16: forward activations of the mixed mode only
NT, NN, TN = 0, 1, 2


def dt(t, split8_ok=False):
    """dtype code of the C ABI: 0 f32, 1 bf16, 2 f16; 3 = an f32-sized activation stored in the split8 layout (Split8 below).
    Only avsr_scale_dropout reads that layout through a dtype code; any other binding that asks for the code of a Split8 tensor
    fails here, loudly, instead of handing its bytes to a kernel as f32."""
    if isinstance(t, Split8):
        if not split8_ok:
            raise TypeError("a split8-layout activation reached an entry point that cannot read it")
        return 3
    return _DT[t.dtype]


def _ptr(t):
    return None if t is None else t.data_ptr()


# raw hipStream_t of torch's current stream: ~0.3 us, against ~5 us for torch.cuda.current_stream(dev).cuda_stream -- with
# ~1500 launches per eager training step that difference is several milliseconds of host time per step
_raw_current_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(*ts):
    L = _lib.lib()
    for t in ts:
        if t is None:
            continue
        if t.is_cuda:
            if L.is_emulator:
                raise _lib.AvsrLibraryError("emulator build cannot take device tensors")
            if _raw_current_stream is not None:
                idx = t.device.index
                return _raw_current_stream(torch.cuda.current_device() if idx is None else idx)
            return torch.cuda.current_stream(t.device).cuda_stream
        if not L.is_emulator:
            raise _lib.AvsrLibraryError(
                "libavsr_hip.so operates on GPU memory only: got a CPU tensor (no CPU fallback exists)"
            )
        return None
    return None


zeros_f32 = lambda shape, device: torch.zeros(shape, dtype=torch.float32, device=device)  # functional.py installs its arena


PROFILE = None  # bench.py: list collecting (entry point, start event, end event, flops) per launch
RECORD = None   # bench.py: (entry points, list) -- the argument tuples of every launch of those entry points
TRACE = None    # tools/pmc_step.py: list collecting (entry point, algorithmic flops, algorithmic bytes) of every launch
PAIR_RECORD = None  # bench.py: list collecting, per paired launch (paired() below), the [(entry point, args, flops, bytes), ...] inside it
_pair_cur = None
_in_pair = 0  # > 0 inside a paired() block, also while pairing itself is switched off by the profiling hooks


def call(name, *args, flops=0.0, nbytes=0.0):
    """flops / nbytes: ALGORITHMIC work of the launch (2 * MACs; every operand read once + every result written once),
    recorded by bench.py's roofline hooks."""
    if TRACE is not None:
        TRACE.append((name, flops, nbytes))
    if RECORD is not None and name in RECORD[0]:
        RECORD[1].append((name, args, flops, nbytes, _in_pair > 0))
    if _pair_cur is not None:
        _pair_cur.append((name, args, flops, nbytes))
    if PROFILE is None:
        return _lib.lib().call(name, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = _lib.lib().call(name, *args)
    e1.record()
    PROFILE.append((name, e0, e1, flops, nbytes))
    return rc


def _nb(*ts):
    """Bytes of the given tensors (None entries skipped)."""
    return float(sum(t.numel() * t.element_size() for t in ts if t is not None))


SPLIT_FAST = os.environ.get("AVSR_SPLIT_FAST", "1") != "0"  # A/B switch: precise-mode forward GEMMs / convs on csrc/gemm_split.hip
PAIR_GEMMS = os.environ.get("AVSR_PAIR_GEMMS", "1") != "0"  # A/B switch for the paired backward GEMMs


@contextlib.contextmanager
def paired():
    """The data-gradient (NT) and weight-gradient (TN) GEMM issued inside the block leave as ONE launch whose grid holds
    the tiles of both (csrc/gemm_pair.hip) -- at M = B*T <= 1600 either one alone cannot fill the 256 CUs.  The two must
    be independent.  Switched off while bench.py's per-launch profiling hooks are installed."""
    global _pair_cur, _in_pair
    if not PAIR_GEMMS or PROFILE is not None or RECORD is not None:
        _in_pair += 1
        try:
            yield
        finally:
            _in_pair -= 1
        return
    L = _lib.lib()
    if PAIR_RECORD is not None:
        _pair_cur = []
    L.call("avsr_gemm_pair_begin")
    try:
        yield
    finally:
        L.call("avsr_gemm_pair_end")
        if PAIR_RECORD is not None and _pair_cur is not None:
            PAIR_RECORD.append(_pair_cur)
        _pair_cur = None


TWIN = None  # functional.py ("hpf" / "mixed" modes): callable(f32 / f16 tensor) -> bf16 twin buffer to fill alongside it, or None


def _twin(y):
    """bf16 twin buffer for an f32 (split-plane forward) or f16 (mixed mode) result `y` when the mode wants one (the bf16
    backward pass reads the twin)."""
    return TWIN(y) if (TWIN is not None and y.dtype in (torch.float32, torch.float16)) else None


def layernorm_fwd(x, gamma, beta, out_dtype, eps=1e-12, twin=False):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    y2 = _twin(y) if twin else None
    if out_dtype == torch.float16:
        call("avsr_layernorm_fwd_h16", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(y2), _ptr(mean), _ptr(rstd), rows, cols, eps,
             _stream(x), nbytes=_nb(x, y, y2))
        return y, mean, rstd
    if y2 is not None:
        call("avsr_layernorm_fwd2", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(y2), _ptr(mean), _ptr(rstd), rows, cols, eps,
             _stream(x), nbytes=_nb(x, y, y2))
        return y, mean, rstd
    call("avsr_layernorm_fwd", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), dt(y), _ptr(mean), _ptr(rstd),
         rows, cols, eps, _stream(x), nbytes=_nb(x, y))
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, gout=None, gsum=None, alpha=1.0, drop_p=0.0,
                  seed=0, seed_dev=None):
    """gout (bf16, same shape as x) / gsum (f32 [cols], zero-initialised): also emit bf16(alpha * dropout(dx)) and its
    column sums -- the backward prologue of the Linear that consumes dx."""
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x)
    call("avsr_layernorm_bwd", _ptr(dy), dt(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dres),
         _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(gout), _ptr(gsum), alpha, drop_p, seed, _ptr(seed_dev), rows, cols,
         _stream(x), nbytes=_nb(dy, x, dres, dx, gout))
    return dx


def gemm(layout, A, lda, B, ldb, M, N, K, C, ldc, *, precise=False, bias=None, act=0, gate=None, ldg=0,
         gate_scale=1.0, drop_p=0.0, seed=0, seed_dev=None, alpha=1.0, alpha_dev=None, resid=None, ldr=0,
         accumulate=False, split_k=1, force_tile=0):
    call("avsr_gemm", layout, _ptr(A), dt(A), lda, _ptr(B), dt(B), ldb, M, N, K, int(precise), _ptr(bias), act,
         _ptr(gate), dt(gate) if gate is not None else 0, ldg, gate_scale, drop_p, seed, _ptr(seed_dev), alpha, _ptr(alpha_dev), _ptr(resid), ldr,
         _ptr(C), dt(C), ldc, int(accumulate), split_k, force_tile, _stream(A), flops=2.0 * M * N * K)
    return C


def attention_fwd(qu, qv, k, v, pos, mask, scale, *, precise=False, drop_p=0.0, seed=0, seed_dev=None):
    """qu/qv/k/v: [B,T,H,64] (possibly strided views with contiguous last two dims); pos: [2T-1, H*64] or None;
    mask: uint8/bool [B,1,Tk] or [B,Tq,Tk] or None.  Returns (out [B,Tq,H*64], lse [B,H,Tq])."""
    B, Tq, H, dk = qu.shape
    Tk = k.shape[1]
    out = torch.empty(B, Tq, H * dk, dtype=qu.dtype, device=qu.device)
    lse = torch.empty(B, H, Tq, dtype=torch.float32, device=qu.device)
    msb = msq = 0
    if mask is not None:
        assert mask.dtype in (torch.uint8, torch.bool) and mask.is_contiguous() and mask.dim() == 3
        msb = mask.shape[1] * mask.shape[2] if mask.shape[0] > 1 else 0  # a batch-1 mask is shared by every sequence
        msq = mask.shape[2] if mask.shape[1] > 1 else 0
    if qu.dtype == torch.float16:  # mixed mode: f16 operands on the transposed-formulation kernel + bf16 twin of the output
        out2 = _twin(out)
        call("avsr_attention_fwd_h16", _ptr(qu), _ptr(qv), _ptr(k), _ptr(v), _ptr(pos), _ptr(mask), msb, msq, _ptr(out), _ptr(out2),
             _ptr(lse), B, H, Tq, Tk, dk, qu.stride(1), k.stride(1), v.stride(1), pos.stride(0) if pos is not None else 0,
             out.stride(1), qu.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, drop_p, seed, _ptr(seed_dev), _stream(qu))
        return out, lse
    out2 = _twin(out) if (precise and out.dtype == torch.float32) else None
    if out2 is not None:  # f32 forward + the bf16 twin of its output in one pass ("hpf" mode)
        call("avsr_attention_fwd2", _ptr(qu), _ptr(qv), _ptr(k), _ptr(v), _ptr(pos), _ptr(mask), msb, msq, _ptr(out), _ptr(out2),
             _ptr(lse), B, H, Tq, Tk, dk, qu.stride(1), k.stride(1), v.stride(1), pos.stride(0) if pos is not None else 0,
             out.stride(1), qu.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, drop_p, seed, _ptr(seed_dev), _stream(qu))
        return out, lse
    call("avsr_attention_fwd", _ptr(qu), _ptr(qv), _ptr(k), _ptr(v), _ptr(pos), dt(qu), int(precise), _ptr(mask),
         msb, msq, _ptr(out), _ptr(lse), B, H, Tq, Tk, dk, qu.stride(1), k.stride(1), v.stride(1),
         pos.stride(0) if pos is not None else 0, out.stride(1), qu.stride(0), k.stride(0), v.stride(0),
         out.stride(0), scale, drop_p, seed, _ptr(seed_dev), _stream(qu))
    return out, lse


def attention_bwd_dq(qu, qv, k, v, pos, mask, out, lse, dout, scale, *, precise=False, drop_p=0.0, seed=0,
                     seed_dev=None, dqu_out=None, dq_sum=None, du=None, dv=None):
    """dq_sum (+ du, dv; relative-position form): the kernel writes dqu + dqv into the [B,Tq,H,64] view dq_sum and adds the
    column sums of dqu / dqv to du / dv (f32 [H*64], zeroed by the caller); dqu / dqv themselves are then not produced."""
    B, Tq, H, dk = qu.shape
    Tk = k.shape[1]
    lds = (Tk + 7) // 8 * 8
    # the kernel addresses dqu / dqv with qu's strides: dqu_out (a [B,Tq,H,64] view, e.g. the q third of a fused
    # d(qkv) buffer) must be laid out like qu
    if dq_sum is not None:
        assert pos is not None and du is not None and dv is not None and dq_sum.shape == qu.shape and dq_sum.stride(2) == dk \
            and dq_sum.stride(3) == 1 and du.dtype == dv.dtype == torch.float32 and du.numel() == dv.numel() == H * dk
        dqu = dqv = None
    else:
        dqu = dqu_out if dqu_out is not None else torch.empty(B, Tq, H, dk, dtype=qu.dtype, device=qu.device)
        dqv = torch.empty(B, Tq, H, dk, dtype=qu.dtype, device=qu.device) if pos is not None else None
    # pad columns [Tk, lds) are never read: the TN loaders mask by the logical width
    pd = torch.empty(B, H, Tq, lds, dtype=qu.dtype, device=qu.device)
    ds = torch.empty(B, H, Tq, lds, dtype=qu.dtype, device=qu.device)
    msb = msq = 0
    if mask is not None:
        msb = mask.shape[1] * mask.shape[2] if mask.shape[0] > 1 else 0  # a batch-1 mask is shared by every sequence
        msq = mask.shape[2] if mask.shape[1] > 1 else 0
    assert dq_sum is not None or (dqu.stride() == qu.stride() and
                                  (dqv is None or (dqv.stride() == qu.stride() and qv.stride() == qu.stride()))), \
        "bwd writes dqu/dqv with qu's strides"
    call("avsr_attention_bwd_dq", _ptr(qu), _ptr(qv), _ptr(k), _ptr(v), _ptr(pos), dt(qu), int(precise), _ptr(mask),
         msb, msq, _ptr(out), _ptr(lse), _ptr(dout), _ptr(dqu), _ptr(dqv), _ptr(pd), _ptr(ds), lds, B, H, Tq, Tk, dk,
         qu.stride(1), k.stride(1), v.stride(1), pos.stride(0) if pos is not None else 0, out.stride(1),
         qu.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, drop_p, seed, _ptr(seed_dev), _ptr(dq_sum),
         dq_sum.stride(1) if dq_sum is not None else 0, dq_sum.stride(0) if dq_sum is not None else 0, _ptr(du), _ptr(dv),
         _stream(qu))
    return dqu, dqv, pd, ds


def gemm_tn_batched(A, lda, sAb, sAh, Bm, ldb, sBb, sBh, C, ldc, sCb, sCh, nb, nh, M, N, K, *, precise=False,
                    accumulate=False, a_skew=False, skew_off=0, skew_lim=0):
    call("avsr_gemm_tn_batched", _ptr(A), dt(A), lda, sAb, sAh, _ptr(Bm), dt(Bm), ldb, sBb, sBh, _ptr(C), dt(C), ldc,
         sCb, sCh, nb, nh, M, N, K, int(precise), int(accumulate), int(a_skew), skew_off, skew_lim, _stream(A))
    return C


def attention_bwd(qu, qv, k, v, pos, mask, out, lse, dout, scale, *, precise=False, drop_p=0.0, seed=0,
                  seed_dev=None, dqu_out=None, dk_out=None, dv_out=None, dpos_out=None, dq_sum=None, du=None, dv_bias=None):
    """Full attention backward.  Returns dqu, dqv (or None), dk, dv, dpos (f32 [2T-1, H*64] or None).
    dqu_out / dk_out / dv_out: optional [B,T,H,64] destination views (slices of a fused d(qkv) buffer).
    dpos_out: optional ZEROED f32 [2T-1, H*64] destination view (a column block of an all-layer buffer)."""
    B, Tq, H, dk = qu.shape
    Tk = k.shape[1]
    D = H * dk
    dqu, dqv, pd, ds = attention_bwd_dq(qu, qv, k, v, pos, mask, out, lse, dout, scale, precise=precise,
                                        drop_p=drop_p, seed=seed, seed_dev=seed_dev, dqu_out=dqu_out, dq_sum=dq_sum, du=du,
                                        dv=dv_bias)
    lds = pd.shape[-1]
    dkk = dk_out if dk_out is not None else torch.empty(B, Tk, H, dk, dtype=qu.dtype, device=qu.device)
    dvv = dv_out if dv_out is not None else torch.empty(B, Tk, H, dk, dtype=qu.dtype, device=qu.device)
    do4 = dout.view(B, Tq, H, dk)
    # dV[b,h] = Pd[b,h]^T dO[b,h] ; dK[b,h] = dS[b,h]^T Qu[b,h] ; dpos += skew(dS[b,h])^T Qv[b,h] -- one launch
    dpos = None
    if pos is not None:  # accumulated into
        dpos = dpos_out if dpos_out is not None else zeros_f32((2 * Tq - 1, D), qu.device)
        assert dpos.dtype == torch.float32 and dpos.shape == (2 * Tq - 1, D) and dpos.stride(1) == 1
    assert qv is None or qv.stride() == qu.stride()
    assert dkk.stride(2) == dk and dvv.stride(2) == dk and do4.stride(2) == dk and qu.stride(2) == dk
    call("avsr_attention_bwd_kv", _ptr(pd), _ptr(ds), lds, _ptr(do4), do4.stride(1), do4.stride(0), _ptr(qu),
         _ptr(qv) if pos is not None else None, qu.stride(1), qu.stride(0), _ptr(dkk), dkk.stride(1), dkk.stride(0),
         _ptr(dvv), dvv.stride(1), dvv.stride(0), _ptr(dpos), dpos.stride(0) if dpos is not None else 0, dt(qu),
         int(precise), B, H, Tq, Tk, dk, _stream(qu))
    return dqu, dqv, dkk, dvv, dpos


def scale_dropout(x, out_dtype, alpha=1.0, drop_p=0.0, seed=0, alpha_dev=None, seed_dev=None, add=None,
                  add_period=0):
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    call("avsr_scale_dropout", _ptr(x), dt(x, split8_ok=True), _ptr(out), dt(out), x.numel(), alpha, _ptr(alpha_dev), drop_p, seed,
         _ptr(seed_dev), _ptr(add), add_period, _stream(x))
    return out


def head_bias_fwd(x, ldx, rows, cols, b1, b2):
    o1 = torch.empty(rows, cols, dtype=x.dtype, device=x.device)
    o2 = torch.empty(rows, cols, dtype=x.dtype, device=x.device)
    t1 = _twin(o1)
    t2 = _twin(o2) if t1 is not None else None
    if x.dtype == torch.float16:
        call("avsr_head_bias_fwd_h16", _ptr(x), ldx, _ptr(b1), _ptr(b2), _ptr(o1), _ptr(o2), _ptr(t1), _ptr(t2), rows, cols, _stream(x))
        return o1, o2
    if t2 is not None:
        call("avsr_head_bias_fwd2", _ptr(x), ldx, _ptr(b1), _ptr(b2), _ptr(o1), _ptr(o2), _ptr(t1), _ptr(t2), rows, cols, _stream(x))
        return o1, o2
    call("avsr_head_bias_fwd", _ptr(x), dt(x), ldx, _ptr(b1), _ptr(b2), _ptr(o1), _ptr(o2), rows, cols, _stream(x))
    return o1, o2


def head_bias_bwd(d1, d2, dq, ldo, db1, db2, rows, cols):
    call("avsr_head_bias_bwd", _ptr(d1), _ptr(d2), dt(d1), _ptr(dq), ldo, _ptr(db1), _ptr(db2), rows, cols,
         _stream(d1))


def colsum_into(d, db, rows, cols):
    """db[c] += sum_r d[r, c]  (bias gradient)."""
    head_bias_bwd(d, None, None, 0, db, None, rows, cols)


def glu_fwd(a, rows, C):
    g = torch.empty(rows, C, dtype=a.dtype, device=a.device)
    call("avsr_glu_fwd", _ptr(a), _ptr(g), dt(a), rows, C, _stream(a))
    return g


def glu_bwd(a, dg, rows, C):
    da = torch.empty(rows, 2 * C, dtype=a.dtype, device=a.device)
    call("avsr_glu_bwd", _ptr(a), _ptr(dg), _ptr(da), dt(a), rows, C, _stream(a))
    return da


def dwconv(x, w, bias, B, T, C, K, flip=False, glu_in=False, glu_a=None):
    """glu_in: x is the pre-GLU tensor [B*T, 2C] (the conv runs on glu(x)); glu_a (flip only): the result is pushed
    through the GLU backward of glu_a = [a | g] -> returns da [B*T, 2C]."""
    y = torch.empty(B, T, 2 * C if glu_a is not None else C, dtype=x.dtype, device=x.device)
    y2 = _twin(y) if (glu_a is None and not flip) else None
    if x.dtype == torch.float16:
        assert glu_a is None and not flip, "f16 depthwise convolution: forward only"
        call("avsr_dwconv_fwd_h16", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(y2), B, T, C, K, int(glu_in), _stream(x),
             nbytes=_nb(x, y, y2))
        return y
    if y2 is not None:  # f32 forward + the bf16 twin of its output ("hpf" mode)
        call("avsr_dwconv_fwd2", _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(y2), B, T, C, K, int(glu_in), _stream(x),
             nbytes=_nb(x, y, y2))
        return y
    call("avsr_dwconv_fwd", _ptr(x), dt(x), _ptr(w), _ptr(bias), _ptr(y), B, T, C, K, int(flip), int(glu_in), _ptr(glu_a),
         _stream(x), nbytes=_nb(x, y, glu_a))
    return y


def dwconv_wgrad(x, dy, dw, db, B, T, C, K, glu_in=False):
    call("avsr_dwconv_wgrad", _ptr(x), _ptr(dy), dt(x), _ptr(dw), _ptr(db), B, T, C, K, int(glu_in), _stream(x),
         nbytes=_nb(x, dy))


def bn_stats(x, rows, C, with_count=False):
    """[3][C] shifted statistics; with_count: returns the flat [3*C + 1] payload {stats, row count} instead."""
    flat = torch.empty(3 * C + (1 if with_count else 0), dtype=torch.float32, device=x.device)
    ws = torch.empty(1024 * 2 * C, dtype=torch.float32, device=x.device)
    call("avsr_bn_stats", _ptr(x), dt(x), _ptr(flat), _ptr(ws), rows, C,
         flat.data_ptr() + 12 * C if with_count else None, _stream(x), nbytes=_nb(x))
    return flat if with_count else flat.view(3, C)


def bn_stats_finalize(x, rows, C, eps, momentum, running_mean, running_var, num_batches_tracked=None):
    """(mean, invstd) of the rows of x + running-stat update, single rank."""
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(1024 * 2 * C, dtype=torch.float32, device=x.device)
    call("avsr_bn_stats_finalize", _ptr(x), dt(x), _ptr(ws), rows, C, eps, momentum, _ptr(mean), _ptr(invstd),
         _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _stream(x), nbytes=_nb(x))
    return mean, invstd


def bn_finalize_parts(part, rows, C, eps, momentum, running_mean, running_var, num_batches_tracked=None):
    """(mean, invstd) + running-stat update from the partial statistics a convolution epilogue left (conv2d_fwd(stats=part))."""
    mean = torch.empty(C, dtype=torch.float32, device=part.device)
    invstd = torch.empty(C, dtype=torch.float32, device=part.device)
    ws = torch.empty(512 * C, dtype=torch.float32, device=part.device)
    call("avsr_bn_finalize_parts", _ptr(part), part.shape[0], C, _ptr(zero_page(part.device)), _ptr(ws), rows, eps, momentum, _ptr(mean),
         _ptr(invstd), _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _stream(part), nbytes=_nb(part))
    return mean, invstd


def bn_stats_parts(part, rows, C):
    """The flat [3 * C + 1] payload {shift = 0, sums, sums of squares, row count} of the cross-rank merge from partial statistics."""
    flat = torch.empty(3 * C + 1, dtype=torch.float32, device=part.device)
    ws = torch.empty(512 * C, dtype=torch.float32, device=part.device)
    call("avsr_bn_stats_parts", _ptr(part), part.shape[0], C, _ptr(zero_page(part.device)), _ptr(ws), _ptr(flat),
         flat.data_ptr() + 12 * C, rows, _stream(part), nbytes=_nb(part))
    return flat


BN_SMALL_MAX_ROWS = 2048  # avsr_bn_small_max_rows()


def bn_small_fwd(x, rows, C, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, act, out_dtype=None):
    """Single-rank BatchNorm + activation of a small [rows, C] activation in ONE launch: (y, mean, invstd).  out_dtype (f32 input
    only): torch.float16 -- the normalised output leaves as f16 (+ its bf16 twin): the next consumer is an MFMA operand."""
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    y = torch.empty(rows, C, dtype=out_dtype or x.dtype, device=x.device)
    y2 = _twin(y)
    if x.dtype == torch.float32 and y.dtype == torch.float16:
        call("avsr_bn_small_fwd2", _ptr(x), rows, C, _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean),
             _ptr(running_var), _ptr(num_batches_tracked), act, _ptr(y), 2, _ptr(y2), _ptr(mean), _ptr(invstd), _stream(x),
             nbytes=_nb(x) + _nb(y) + _nb(y2))
        return y, mean, invstd
    assert y.dtype == x.dtype
    if x.dtype == torch.float16:
        call("avsr_bn_small_fwd_h16", _ptr(x), rows, C, _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean),
             _ptr(running_var), _ptr(num_batches_tracked), act, _ptr(y), _ptr(y2), _ptr(mean), _ptr(invstd), _stream(x),
             nbytes=2.0 * _nb(x) + _nb(y2))
        return y, mean, invstd
    if y2 is not None:
        call("avsr_bn_small_fwd2", _ptr(x), rows, C, _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean),
             _ptr(running_var), _ptr(num_batches_tracked), act, _ptr(y), 0, _ptr(y2), _ptr(mean), _ptr(invstd), _stream(x),
             nbytes=2.0 * _nb(x) + _nb(y2))
        return y, mean, invstd
    call("avsr_bn_small_fwd", _ptr(x), dt(x), rows, C, _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean),
         _ptr(running_var), _ptr(num_batches_tracked), act, _ptr(y), _ptr(mean), _ptr(invstd), _stream(x),
         nbytes=2.0 * _nb(x))
    return y, mean, invstd


def bn_small_bwd(x, dy, rows, C, mean, invstd, gamma, beta, act):
    """Backward of bn_small_fwd in ONE launch: (dx, dgamma, dbeta)."""
    dx = torch.empty(rows, C, dtype=x.dtype, device=x.device)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    call("avsr_bn_small_bwd", _ptr(x), _ptr(dy), dt(x), rows, C, _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), act,
         _ptr(dx), _ptr(dgamma), _ptr(dbeta), _stream(x), nbytes=3.0 * _nb(x))
    return dx, dgamma, dbeta


def convmod_dwbn_fwd(a, wdw, bdw, B, T, C, K, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                     out_dtype=None):
    """GLU -> depthwise conv -> BatchNorm (batch statistics) -> SiLU of a [B*T, 2C] in ONE launch (csrc/convmod_fused.hip):
    (s, c_saved, mean, invstd); c_saved is the depthwise output the backward pass reads -- its bf16 twin when the mode keeps twins
    (the f32 tensor itself is then never written), the tensor in a's dtype otherwise."""
    rows = B * T
    s = torch.empty(rows, C, dtype=out_dtype or a.dtype, device=a.device)
    s2 = _twin(s)
    c2 = torch.empty(rows, C, dtype=torch.bfloat16, device=a.device) if (s2 is not None and a.dtype == torch.float32) else None
    c = torch.empty(rows, C, dtype=a.dtype, device=a.device) if c2 is None else None
    mean = torch.empty(C, dtype=torch.float32, device=a.device)
    invstd = torch.empty(C, dtype=torch.float32, device=a.device)
    call("avsr_convmod_dwbn_fwd", _ptr(a), dt(a), _ptr(wdw), _ptr(bdw), B, T, C, K, _ptr(gamma), _ptr(beta), eps, momentum,
         _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _ptr(c), _ptr(c2), _ptr(s), dt(s), _ptr(s2), _ptr(mean),
         _ptr(invstd), _stream(a), nbytes=_nb(a, c, c2, s, s2))
    return s, (c2 if c2 is not None else c), mean, invstd


def convmod_dwbn_bwd(a, c, ds, mean, invstd, gamma, beta, wdw, B, T, C, K, dwdw, dbdw):
    """Backward of convmod_dwbn_fwd in ONE launch: (da [B*T, 2C], dgamma, dbeta); dwdw [C, K] / dbdw [C] are added to."""
    assert a.dtype == c.dtype == ds.dtype, (a.dtype, c.dtype, ds.dtype)
    rows = B * T
    da = torch.empty(rows, 2 * C, dtype=a.dtype, device=a.device)
    dgamma = torch.empty(C, dtype=torch.float32, device=a.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=a.device)
    call("avsr_convmod_dwbn_bwd", _ptr(a), _ptr(c), _ptr(ds), dt(a), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(wdw),
         B, T, C, K, _ptr(da), _ptr(dwdw), _ptr(dbdw), _ptr(dgamma), _ptr(dbeta), _stream(a), nbytes=_nb(a, c, ds, da))
    return da, dgamma, dbeta


def bn_finalize(stats, counts, world, C, eps, momentum, running_mean, running_var, num_batches_tracked=None,
                stats_stride=0, counts_stride=0, n_total=None):
    """counts: tensor or raw device address of the first count."""
    mean = torch.empty(C, dtype=torch.float32, device=stats.device)
    invstd = torch.empty(C, dtype=torch.float32, device=stats.device)
    call("avsr_bn_finalize", _ptr(stats), counts if isinstance(counts, int) else _ptr(counts), world, C, stats_stride,
         counts_stride, eps, momentum, _ptr(mean), _ptr(invstd), _ptr(running_mean), _ptr(running_var),
         _ptr(num_batches_tracked), _ptr(n_total), _stream(stats))
    return mean, invstd


def bn_eval_params(running_mean, running_var, eps):
    C = running_mean.numel()
    mean = torch.empty_like(running_mean)
    invstd = torch.empty_like(running_mean)
    call("avsr_bn_eval_params", _ptr(running_mean), _ptr(running_var), eps, C, _ptr(mean), _ptr(invstd),
         _stream(running_mean))
    return mean, invstd


def bn_act_fwd(x, add, mean, invstd, gamma, beta, rows, C, act, out_split8=False):
    """out_split8 (f32 input only): the output leaves in the split8 layout (a Split8-tagged f32-sized tensor) -- what the
    split-plane convolution stages without a conversion pass; `add` may be Split8-tagged itself."""
    y = torch.empty_like(x)
    if isinstance(y, Split8):
        y = y.as_subclass(torch.Tensor)
    add_s8 = isinstance(add, Split8)
    assert not (out_split8 or add_s8) or (x.dtype == torch.float32 and not isinstance(x, Split8))
    if x.dtype == torch.float16:
        assert add is None or add.dtype == torch.float16
        y2 = _twin(y)
        call("avsr_bn_act_fwd_h16", _ptr(x), _ptr(add), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(y2), rows, C,
             act, _stream(x), nbytes=_nb(x, add, y, y2))
        return y
    y2 = _twin(y) if (add is None or add.dtype == torch.float32) else None
    if y2 is not None or out_split8 or add_s8:
        call("avsr_bn_act_fwd2", _ptr(x), _ptr(add), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(y2), rows, C,
             act, int(out_split8) | (2 if add_s8 else 0), _stream(x), nbytes=_nb(x, add, y, y2))
        return y.as_subclass(Split8) if out_split8 else y
    call("avsr_bn_act_fwd", _ptr(x), _ptr(add), dt(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(y),
         rows, C, act, _stream(x), nbytes=_nb(x, add, y))
    return y


def bn_act_pool_fwd(x, mean, invstd, gamma, beta, N, H, W, C, K, S, P, act, want_xsel=False, out_split8=False):
    """maxpool(act(bn(x))) without the full-resolution activation; returns (y, idx) like maxpool2d_fwd -- and with
    want_xsel (3x3 / stride 2 / pad 1 only) also xsel, the raw x at every arg-max."""
    OH, OW = conv_out(H, K, S, P), conv_out(W, K, S, P)
    y = torch.empty(N, OH, OW, C, dtype=x.dtype, device=x.device)
    idx = torch.empty(N, OH, OW, C, dtype=torch.uint8, device=x.device)
    if x.dtype == torch.float32 and (K, S, P) == (3, 2, 1) and want_xsel:
        y2 = _twin(y)
        if y2 is not None or out_split8:  # hpf / mixed modes: the pooled output's bf16 twin and the arg-max inputs in bf16, from the same pass
            xsel = torch.empty(N, OH, OW, C, dtype=torch.bfloat16, device=x.device)
            call("avsr_bn_act_pool3_fwd2", _ptr(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(y2), _ptr(idx),
                 _ptr(xsel), N, H, W, C, act, int(out_split8), _stream(x), nbytes=_nb(x, y, y2, idx, xsel))
            return (y.as_subclass(Split8) if out_split8 else y), idx, xsel
    assert not out_split8, "split8 output: the fused 3x3 / stride-2 stem pass with a producer-side twin only"
    xsel = torch.empty_like(y) if want_xsel else None
    call("avsr_bn_act_pool_fwd", _ptr(x), dt(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(idx),
         _ptr(xsel), N, H, W, C, K, S, P, act, _stream(x), nbytes=_nb(x, y, idx, xsel))
    return (y, idx, xsel) if want_xsel else (y, idx)


def bn_pool_bwd_reduce(x, dpool, idx, mean, invstd, gamma, beta, N, H, W, C, K, S, P, act):
    sums = torch.empty(2, C, dtype=torch.float32, device=x.device)
    ws = torch.empty(1024 * 2 * C, dtype=torch.float32, device=x.device)
    call("avsr_bn_pool_bwd_reduce", _ptr(x), _ptr(dpool), _ptr(idx), dt(x), _ptr(mean), _ptr(invstd), _ptr(gamma),
         _ptr(beta), _ptr(sums), _ptr(ws), N, H, W, C, K, S, P, act, _stream(x), nbytes=_nb(x, dpool, idx))
    return sums


def bn_pool_bwd_apply(x, dpool, idx, mean, invstd, gamma, beta, sums, inv_n, N, H, W, C, K, S, P, act, n_dev=None):
    dx = torch.empty_like(x)
    call("avsr_bn_pool_bwd_apply", _ptr(x), _ptr(dpool), _ptr(idx), dt(x), _ptr(mean), _ptr(invstd), _ptr(gamma),
         _ptr(beta), _ptr(sums), inv_n, _ptr(n_dev), _ptr(dx), N, H, W, C, K, S, P, act, _stream(x),
         nbytes=_nb(x, dpool, idx, dx))
    return dx


def bn_bwd_reduce(x, dy, add, mean, invstd, gamma, beta, rows, C, act):
    sums = torch.empty(2, C, dtype=torch.float32, device=x.device)
    ws = torch.empty(1024 * 2 * C, dtype=torch.float32, device=x.device)
    call("avsr_bn_bwd_reduce", _ptr(x), _ptr(dy), _ptr(add), dt(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
         _ptr(sums), _ptr(ws), rows, C, act, _stream(x), nbytes=_nb(x, dy, add))
    return sums


def bn_bwd_apply(x, dy, add, mean, invstd, gamma, beta, sums, inv_n, rows, C, act, want_dadd, n_dev=None):
    dx = torch.empty_like(x)
    dadd = torch.empty_like(x) if want_dadd else None
    call("avsr_bn_bwd_apply", _ptr(x), _ptr(dy), _ptr(add), dt(x), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta),
         _ptr(sums), inv_n, _ptr(n_dev), _ptr(dx), _ptr(dadd), rows, C, act, _stream(x), nbytes=_nb(x, dy, add, dx, dadd))
    return dx, dadd


def ctc_loss(logits, ld, labels, in_lens, B, T, V, want_grad=True, ignore_id=-1):
    """logits: [B*T, ld] (f32/bf16); labels int64 [B, Lmax]; in_lens int64 [B].
    Returns nll [B] f32 (inf = infeasible) and grad [B*T, ld] (same dtype) or None."""
    Lmax = labels.shape[1]
    ws_bytes = call("avsr_ctc_workspace_bytes", B, T, Lmax)
    ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device=logits.device)
    nll = torch.empty(B, dtype=torch.float32, device=logits.device)
    grad = torch.empty(B * T, ld, dtype=logits.dtype, device=logits.device) if want_grad else None  # (the kernel writes the pad columns too)
    call("avsr_ctc_loss", _ptr(logits), dt(logits), ld, _ptr(labels), Lmax, ignore_id, _ptr(in_lens), _ptr(nll),
         _ptr(grad), ld, _ptr(ws), B, T, V, _stream(logits))
    return nll, grad


def ce_smooth(logits, ld, target, V, smoothing, want_grad=True, ignore_id=-1):
    rows = target.numel()
    row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
    row_hit = torch.empty(rows, dtype=torch.float32, device=logits.device)
    grad = torch.empty(rows, ld, dtype=logits.dtype, device=logits.device) if want_grad else None  # (the kernel writes the pad columns too)
    call("avsr_ce_smooth", _ptr(logits), dt(logits), ld, _ptr(target), ignore_id, V, smoothing, _ptr(row_loss),
         _ptr(row_hit), _ptr(grad), ld, rows, _stream(logits))
    return row_loss, row_hit, grad


def sum_scale(a, scale, finite_only=False):
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    call("avsr_sum_scale", _ptr(a), a.numel(), scale, _ptr(out), int(finite_only), _stream(a))
    return out


def sum_finite_scale(a, scale):
    return sum_scale(a, scale, finite_only=True)


def prepare_targets(ys_pad, sos, eos, ignore_id, want_mask=True):
    """(ys_in, ys_out [B, L+1] int64, mask [B, L+1, L+1] bool or None, n_tokens [1] int64) -- csrc/loss.hip
    prepare_targets_kernel: add_sos_eos + target_mask of the reference with the static width L + 1, one launch."""
    y = ys_pad.reshape(ys_pad.shape[0], -1).to(torch.int64).contiguous()
    B, L = y.shape
    ys_in = torch.empty(B, L + 1, dtype=torch.int64, device=y.device)
    ys_out = torch.empty(B, L + 1, dtype=torch.int64, device=y.device)
    mask = torch.empty(B, L + 1, L + 1, dtype=torch.bool, device=y.device) if want_mask else None
    n_tok = torch.empty(1, dtype=torch.int64, device=y.device)
    call("avsr_prepare_targets", _ptr(y), B, L, int(sos), int(eos), int(ignore_id), _ptr(ys_in), _ptr(ys_out), _ptr(mask),
         _ptr(n_tok), _stream(y))
    return ys_in, ys_out, mask, n_tok


def embed_fwd(ids, table, pe, L, scale, drop_p=0.0, seed=0, seed_dev=None):
    rows, D = ids.numel(), table.shape[1]
    out = torch.empty(*ids.shape, D, dtype=torch.float32, device=table.device)
    call("avsr_embed_fwd", _ptr(ids), _ptr(table), _ptr(pe), _ptr(out), rows, L, D, scale, drop_p, seed, _ptr(seed_dev),
         _stream(table))
    return out


def embed_bwd(ids, dout, dtable, scale, drop_p=0.0, seed=0, seed_dev=None):
    rows, D = ids.numel(), dtable.shape[1]
    call("avsr_embed_bwd", _ptr(ids), _ptr(dout), _ptr(dtable), rows, D, scale, drop_p, seed, _ptr(seed_dev),
         _stream(dout))


def log_softmax(x, ld, rows, V):
    out = torch.zeros(rows, ld, dtype=torch.float32, device=x.device)  # x may be a [rows, V] view of a pitched buffer
    ws = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("avsr_log_softmax", _ptr(x), ld, _ptr(ws), _ptr(out), rows, V, _stream(x))
    return out


# ---------------------------------------------------------------------------------------------- convolutions
def conv_weight_permute(w, out_dtype, to_dgrad=False, ld_out=None):
    """torch conv weight [Cout, Cin, *taps] (f32) -> [Cout][taps][Cin] (or [Cin][taps][Cout] for the data gradient)."""
    Cout, Cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    a, b = (Cin, Cout) if to_dgrad else (Cout, Cin)
    ld = ld_out or taps * b
    alloc = torch.zeros if ld != taps * b else torch.empty
    out = alloc(a, ld, dtype=out_dtype, device=w.device)
    # (this entry point's dtype code 2 is the split8 layout; IEEE half is 3)
    call("avsr_conv_weight_permute", _ptr(w), _ptr(out), 3 if out_dtype == torch.float16 else dt(out), Cout, Cin, taps,
         int(to_dgrad), ld, _stream(w))
    return out


class Split8(torch.Tensor):
    """Marker type of a buffer in the split8 layout (csrc/gemm_split.hip): an f32-shaped tensor whose every group of 8
    consecutive elements holds 8 hi bf16 + 8 lo bf16 of the values it replaces.  Only the precise-mode GEMM / convolution
    entry points read it (as their pre-split B operand)."""


def conv_weight_permute_split(w, to_dgrad=False, out=None):
    """conv_weight_permute into the split8 layout ([Cout][taps*Cin] f32-shaped buffer)."""
    Cout, Cin = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()
    a, b = (Cin, Cout) if to_dgrad else (Cout, Cin)
    if out is None:
        out = torch.empty(a, taps * b, dtype=torch.float32, device=w.device).as_subclass(Split8)
    call("avsr_conv_weight_permute", _ptr(w), _ptr(out), 2, Cout, Cin, taps, int(to_dgrad), taps * b, _stream(w))
    return out


def split_pack(src, out=None):
    """f32 tensor (numel % 8 == 0, contiguous) -> the same shape in the split8 layout."""
    if out is None:
        out = torch.empty(src.shape, dtype=torch.float32, device=src.device).as_subclass(Split8)
    call("avsr_split_pack", _ptr(src), _ptr(out), src.numel(), _stream(src), nbytes=8.0 * src.numel())
    return out


def multi_split_pack(table_dev, n, blocks):
    call("avsr_multi_split_pack", _ptr(table_dev), n, blocks, _stream(table_dev))


def conv_weight_unpermute(dwp, shape):
    Cout, Cin = shape[0], shape[1]
    taps = 1
    for s in shape[2:]:
        taps *= s
    dw = torch.empty(shape, dtype=torch.float32, device=dwp.device)
    call("avsr_conv_weight_unpermute", _ptr(dwp), _ptr(dw), Cout, Cin, taps, _stream(dwp))
    return dw


def conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


_zero_pages = {}


def zero_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = _zero_pages[device] = torch.zeros(1024, dtype=torch.float32, device=device)  # (>= the widest BatchNorm: its zero shift row)
    return z


def apply_env_tuning():
    """AVSR_TUNE="knob=value,knob=value": tuning knobs for A/B runs of unmodified entry points (bench.py, tests)."""
    spec = os.environ.get("AVSR_TUNE", "")
    for item in filter(None, spec.split(",")):
        k, v = item.split("=")
        call("avsr_tune", int(k), int(v))


def tune(knob, value):
    """Process-wide tuning knob of the tuned kernels (benchmarks): see avsr_tune in include/avsr_hip.h."""
    call("avsr_tune", int(knob), int(value))


def bn_stat_tiles(rows):
    """Rows of the partial-statistics buffer of avsr_conv2d_f32s_stats: one per 128-row output tile."""
    return (rows + 127) // 128


def conv2d_takes_stats(x, wp, Cin, KH, KW, precise):
    """Will conv2d_fwd run on the split-plane kernel, whose epilogue can leave the BatchNorm statistics of the output behind?"""
    return bool(precise and SPLIT_FAST and x.dtype == torch.float32 and wp.dtype == torch.float32 and Cin % 64 == 0 and KH * KW <= 32)


def conv2d_fwd(x, wp, N, H, W, Cin, Cout, KH, KW, stride, ph, pw, precise, stats=None, wp_planes=1):
    """stats (conv2d_takes_stats only): [bn_stat_tiles(rows)][2][Cout] f32 (uninitialised) -- receives per 128-row tile the
    per-column sums / sums of squares of y."""
    OH, OW = conv_out(H, KH, stride, ph), conv_out(W, KW, stride, pw)
    y = torch.empty(N, OH, OW, Cout, dtype=x.dtype, device=x.device)
    assert stats is None or conv2d_takes_stats(x, wp, Cin, KH, KW, precise)
    if not precise and x.dtype == torch.bfloat16 and wp.dtype == torch.bfloat16 and Cin % 64 == 0 and stride <= 2:
        call("avsr_conv2d_bf16", 0, _ptr(x), _ptr(wp), None, _ptr(y), _ptr(zero_page(x.device)), N, H, W, Cin, Cout, KH,
             KW, stride, ph, pw, _stream(x), flops=2.0 * N * OH * OW * Cout * KH * KW * Cin,
             nbytes=_nb(x, wp) + 2.0 * N * OH * OW * Cout)
        return y
    if x.dtype == torch.float16:  # mixed mode, f16 component: the tiled kernel on IEEE-half operands + the bf16 twin of the result
        assert wp.dtype == torch.float16 and Cin % 64 == 0 and stride <= 2 and not precise
        # wp: dense [Cout][taps*Cin], or the two-plane image [Cout][2][taps*Cin] (functional._w_conv_h16) with wp_lo = its lo rows
        # when the component asks for exact weights ("f16x2")
        two = wp.dim() == 3
        hi, lo = (wp[:, 0], wp[:, 1] if wp_planes == 2 else None) if two else (wp, None)
        call("avsr_conv2d_h16", _ptr(x), _ptr(hi), _ptr(lo), hi.stride(0) if two else 0, _ptr(y), _ptr(_twin(y)), _ptr(zero_page(x.device)),
             N, H, W, Cin, Cout, KH, KW, stride, ph, pw, _stream(x), flops=2.0 * N * OH * OW * Cout * KH * KW * Cin,
             nbytes=_nb(x) + (2.0 if lo is not None else 1.0) * 2.0 * Cout * KH * KW * Cin + 4.0 * N * OH * OW * Cout)
        return y
    if precise and SPLIT_FAST and x.dtype == torch.float32 and wp.dtype == torch.float32 and Cin % 64 == 0 and KH * KW <= 32:
        # x in the split8 layout (its producer wrote it that way): the kernel variants that stage A without a conversion pass
        tile = 0
        if isinstance(x, Split8):
            assert isinstance(wp, Split8), "a pre-split activation needs the pre-split filter"
            tile = 24 if Cout >= 128 else 23
        if stats is not None:
            call("avsr_conv2d_f32s_stats", _ptr(x), _ptr(wp), _ptr(y), _ptr(zero_page(x.device)), N, H, W, Cin, Cout, KH, KW, stride,
                 ph, pw, tile, int(isinstance(wp, Split8)), _ptr(_twin(y)), _ptr(stats), stats.shape[0], _stream(x),
                 flops=2.0 * N * OH * OW * Cout * KH * KW * Cin, nbytes=_nb(x, wp, y))
            return y
        call("avsr_conv2d_f32s", _ptr(x), _ptr(wp), _ptr(y), _ptr(zero_page(x.device)), N, H, W, Cin, Cout, KH, KW, stride,
             ph, pw, tile, int(isinstance(wp, Split8)), _ptr(_twin(y)), _stream(x),
             flops=2.0 * N * OH * OW * Cout * KH * KW * Cin, nbytes=_nb(x, wp, y))
        return y
    assert not isinstance(x, Split8), "a pre-split activation reached a convolution path that cannot read it"
    call("avsr_conv2d_fwd", _ptr(x), dt(x), _ptr(wp), dt(wp), _ptr(y), N, H, W, Cin, Cout, KH, KW, stride, ph, pw,
         int(precise), _stream(x), flops=2.0 * N * OH * OW * Cout * KH * KW * Cin)
    return y


def conv2d_dgrad(dy, wpd, resid, N, H, W, Cin, Cout, KH, KW, stride, ph, pw, precise):
    dx = torch.empty(N, H, W, Cin, dtype=dy.dtype, device=dy.device)
    if not precise and dy.dtype == torch.bfloat16 and wpd.dtype == torch.bfloat16 and Cout % 64 == 0 and stride <= 2:
        call("avsr_conv2d_bf16", 1, _ptr(dy), _ptr(wpd), _ptr(resid), _ptr(dx), _ptr(zero_page(dy.device)), N, H, W, Cin,
             Cout, KH, KW, stride, ph, pw, _stream(dy), flops=2.0 * N * H * W * Cin * KH * KW * Cout,
             nbytes=_nb(dy, wpd, resid) + 2.0 * N * H * W * Cin)
        return dx
    call("avsr_conv2d_dgrad", _ptr(dy), dt(dy), _ptr(wpd), dt(wpd), _ptr(resid), _ptr(dx), N, H, W, Cin, Cout, KH, KW,
         stride, ph, pw, int(precise), _stream(dy), flops=2.0 * N * H * W * Cin * KH * KW * Cout)
    return dx


def conv2d_wgrad(dy, x, N, H, W, Cin, Cout, KH, KW, stride, ph, pw, precise, torch_layout=False):
    """Weight gradient.  Returns [Cout, KH*KW*Cin] (permuted, [Cout][KH][KW][Cin]) -- or, with torch_layout=True, the
    gradient already in the parameter's own [Cout, Cin, KH, KW] layout."""
    OH, OW = conv_out(H, KH, stride, ph), conv_out(W, KW, stride, pw)
    bf = not precise and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
    if bf and KH == 3 and KW == 3 and ph == 1 and pw == 1 and stride <= 2 and Cin % 64 == 0 and Cout % 64 == 0 \
            and (OW - 1) * stride + 3 <= 64:
        dwp = torch.empty((Cout, Cin, KH, KW) if torch_layout else (Cout, KH * KW * Cin), dtype=torch.float32, device=x.device)
        nws = call("avsr_conv3x3_wgrad_workspace_bytes", N, H, W, Cin, Cout, stride)
        ws = torch.empty(nws // 4, dtype=torch.float32, device=x.device)
        call("avsr_conv3x3_wgrad_bf16", _ptr(dy), _ptr(x), _ptr(dwp), _ptr(zero_page(x.device)), _ptr(ws), nws, N, H, W, Cin,
             Cout, stride, int(torch_layout), _stream(x), flops=2.0 * N * OH * OW * Cout * KH * KW * Cin,
             nbytes=_nb(dy, x, dwp))
        return dwp
    if torch_layout:
        return conv_weight_unpermute(conv2d_wgrad(dy, x, N, H, W, Cin, Cout, KH, KW, stride, ph, pw, precise), (Cout, Cin, KH, KW))
    dwp = zeros_f32((Cout, KH * KW * Cin), x.device)  # (the zero-scratch arena: no fill launch per weight gradient)
    if bf and Cin % 64 == 0 and Cout % 8 == 0:
        call("avsr_conv2d_wgrad_bf16", _ptr(dy), _ptr(x), _ptr(dwp), _ptr(zero_page(x.device)), N, H, W, Cin, Cout, KH,
             KW, stride, ph, pw, _stream(x), flops=2.0 * N * OH * OW * Cout * KH * KW * Cin, nbytes=_nb(dy, x, dwp))
        return dwp
    call("avsr_conv2d_wgrad", _ptr(dy), _ptr(x), dt(x), _ptr(dwp), N, H, W, Cin, Cout, KH, KW, stride, ph, pw,
         int(precise), _stream(x), flops=2.0 * N * OH * OW * Cout * KH * KW * Cin)
    return dwp


def conv_stem_fwd(x, wp, ldw, out_dtype, B, T, H, W, Cout, KT, KH, KW, stride, pt, ph, pw, precise):
    OH, OW = conv_out(H, KH, stride, ph), conv_out(W, KW, stride, pw)
    y = torch.empty(B * T, OH, OW, Cout, dtype=out_dtype, device=x.device)
    call("avsr_conv_stem_fwd", _ptr(x), _ptr(wp), dt(wp), ldw, _ptr(y), dt(y), B, T, H, W, Cout, KT, KH, KW, stride, pt,
         ph, pw, int(precise), _stream(x), flops=2.0 * B * T * OH * OW * Cout * KT * KH * KW)
    return y


def conv_stem_wgrad(dy, x, B, T, H, W, Cout, KT, KH, KW, stride, pt, ph, pw, precise):
    OH, OW = conv_out(H, KH, stride, ph), conv_out(W, KW, stride, pw)
    dw = torch.zeros(Cout, KT * KH * KW, dtype=torch.float32, device=x.device)
    call("avsr_conv_stem_wgrad", _ptr(dy), dt(dy), _ptr(x), _ptr(dw), B, T, H, W, Cout, KT, KH, KW, stride, pt, ph, pw,
         int(precise), _stream(x), flops=2.0 * B * T * OH * OW * Cout * KT * KH * KW)
    return dw


def maxpool2d_fwd(x, N, H, W, C, K, S, P):
    OH, OW = conv_out(H, K, S, P), conv_out(W, K, S, P)
    y = torch.empty(N, OH, OW, C, dtype=x.dtype, device=x.device)
    idx = torch.empty(N, OH, OW, C, dtype=torch.uint8, device=x.device)
    call("avsr_maxpool2d_fwd", _ptr(x), _ptr(y), _ptr(idx), dt(x), N, H, W, C, K, S, P, _stream(x), nbytes=_nb(x, y, idx))
    return y, idx


def maxpool2d_bwd(idx, dy, N, H, W, C, K, S, P):
    dx = torch.empty(N, H, W, C, dtype=dy.dtype, device=dy.device)
    call("avsr_maxpool2d_bwd", _ptr(idx), _ptr(dy), _ptr(dx), dt(dy), N, H, W, C, K, S, P, _stream(dy), nbytes=_nb(idx, dy, dx))
    return dx


def avgpool_fwd(x, groups, win, C):
    y = torch.empty(groups, C, dtype=torch.float32, device=x.device)
    call("avsr_avgpool_fwd", _ptr(x), dt(x), _ptr(y), groups, win, C, _stream(x))
    return y


def avgpool_bwd(dy, out_dtype, groups, win, C):
    dx = torch.empty(groups * win, C, dtype=out_dtype, device=dy.device)
    call("avsr_avgpool_bwd", _ptr(dy), _ptr(dx), dt(dx), groups, win, C, _stream(dy))
    return dx


def gemm_bf16_nt(A, lda, B, ldb, M, N, K, C, ldc, *, bias=None, act=0, gate=None, ldg=0, gate_scale=1.0, drop_p=0.0,
                 seed=0, seed_dev=None, alpha=1.0, alpha_dev=None, resid=None, ldr=0, accumulate=False, split_k=1,
                 tile=0, colsum=None):
    call("avsr_gemm_bf16_nt", _ptr(A), lda, _ptr(B), ldb, M, N, K, _ptr(bias), act, _ptr(gate),
         dt(gate) if gate is not None else 0, ldg, gate_scale, drop_p, seed, _ptr(seed_dev), alpha, _ptr(alpha_dev),
         _ptr(resid), dt(resid) if resid is not None else 0, ldr, _ptr(C), dt(C), ldc, int(accumulate), split_k, tile,
         _ptr(colsum), _stream(A), flops=2.0 * M * N * K,
         nbytes=2.0 * (M * K + N * K) + float(M * N * C.element_size()) + (float(M * N * resid.element_size()) if resid is not None else 0.0)
         + (2.0 * M * N if gate is not None else 0.0))
    return C


def gemm_f32s_nt(A, lda, B, ldb, M, N, K, C, ldc, *, bias=None, act=0, gate=None, ldg=0, gate_scale=1.0, drop_p=0.0,
                 seed=0, seed_dev=None, alpha=1.0, alpha_dev=None, resid=None, ldr=0, accumulate=False, split_k=1,
                 tile=0, colsum=None, twin=False):
    """Precise-mode NT GEMM on the LDS-DMA ring (csrc/gemm_split.hip): f32 operands, three bf16 MFMAs per product."""
    # hpf mode: a dense f32 activation output also leaves as its bf16 twin (same pitch), for the backward pass
    c2 = _twin(C) if (twin and not accumulate and C.dtype == torch.float32 and C.dim() == 2 and C.stride(0) == ldc and C.is_contiguous()) else None
    call("avsr_gemm_f32s_nt", _ptr(A), lda, _ptr(B), ldb, M, N, K, _ptr(bias), act, _ptr(gate),
         dt(gate) if gate is not None else 0, ldg, gate_scale, drop_p, seed, _ptr(seed_dev), alpha, _ptr(alpha_dev),
         _ptr(resid), dt(resid) if resid is not None else 0, ldr, _ptr(C), dt(C), ldc, int(accumulate), split_k, tile,
         _ptr(colsum), int(isinstance(B, Split8)), _ptr(c2), ldc, _stream(A), flops=2.0 * M * N * K,
         nbytes=4.0 * (M * K + N * K) + float(M * N * C.element_size()) + (float(M * N * resid.element_size()) if resid is not None else 0.0))
    return C


def gemm_h16_nt(A, lda, B, ldb, M, N, K, C, ldc, *, bias=None, act=0, drop_p=0.0, seed=0, seed_dev=None, alpha=1.0,
                resid=None, ldr=0, tile=0, twin=False, B_lo=None):
    """Mixed-mode forward NT GEMM (csrc/gemm_fast.hip, F16 = 1): A and B IEEE half, C f32 / bf16 / f16; twin: a dense f16 / f32
    activation output also leaves as its bf16 twin (same pitch) for the backward pass.  B_lo: the scaled lo plane of the weight
    (same pitch) -- two MFMAs per product, exact weights."""
    assert A.dtype == torch.float16 and B.dtype == torch.float16 and (B_lo is None or B_lo.dtype == torch.float16)
    c2 = _twin(C) if (twin and C.dtype in (torch.float16, torch.float32) and C.dim() == 2 and C.stride(0) == ldc and C.is_contiguous()) else None
    call("avsr_gemm_h16_nt", _ptr(A), lda, _ptr(B), _ptr(B_lo), ldb, M, N, K, _ptr(bias), act, drop_p, seed, _ptr(seed_dev), alpha,
         _ptr(resid), dt(resid) if resid is not None else 0, ldr, _ptr(C), dt(C), ldc, tile, _ptr(c2), ldc, _stream(A),
         flops=2.0 * M * N * K,
         nbytes=2.0 * (M * K + (2 if B_lo is not None else 1) * N * K) + float(M * N * C.element_size())
         + (float(M * N * resid.element_size()) if resid is not None else 0.0) + (2.0 * M * N if c2 is not None else 0.0))
    return C


def transpose_cast(src, R, Ccols, ld_src=None, pad_to=64):
    """bf16 [Ccols, ldd] = src[R, Ccols]^T with ldd = R rounded up to `pad_to` (zero tail)."""
    ldd = (R + pad_to - 1) // pad_to * pad_to
    dst = torch.empty(Ccols, ldd, dtype=torch.bfloat16, device=src.device)
    call("avsr_transpose_cast", _ptr(src), dt(src), ld_src or Ccols, _ptr(dst), ldd, R, Ccols, _stream(src))
    return dst


def cast_transpose_colsum(src, R, Ccols, *, ld_src=None, want_dst=False, want_T=True, colsum=None, alpha=1.0,
                          alpha_dev=None, drop_p=0.0, seed=0, seed_dev=None):
    """One pass over src [R, Ccols]: returns (dst bf16 [R, Ccols] or None, dstT bf16 [Ccols, R->64-padded] or None);
    colsum (f32 [Ccols], accumulated into) optional."""
    dst = torch.empty(R, Ccols, dtype=torch.bfloat16, device=src.device) if want_dst else None
    ldd = (R + 63) // 64 * 64
    dstT = torch.empty(Ccols, ldd, dtype=torch.bfloat16, device=src.device) if want_T else None
    call("avsr_cast_transpose_colsum", _ptr(src), dt(src), ld_src or Ccols, _ptr(dst), _ptr(dstT), ldd, _ptr(colsum), R,
         Ccols, alpha, _ptr(alpha_dev), drop_p, seed, _ptr(seed_dev), _stream(src))
    return dst, dstT


def transpose_cast_into(src, dst):
    R, Ccols = src.shape
    call("avsr_transpose_cast", _ptr(src), dt(src), Ccols, _ptr(dst), dst.shape[1], R, Ccols, _stream(src))


def cast_into(src, dst):
    call("avsr_scale_dropout", _ptr(src), dt(src), _ptr(dst), dt(dst), src.numel(), 1.0, None, 0.0, 0, None, None, 0,
         _stream(src))


def weight_permute_blocks(Cout, Cin, to_dgrad):
    return int(call("avsr_weight_permute_blocks", Cout, Cin, int(to_dgrad)))


def multi_weight_permute(table_dev, n, blocks, max_taps):
    call("avsr_multi_weight_permute", _ptr(table_dev), n, blocks, max_taps, _stream(table_dev))


def multi_cast_transpose(table_dev, n, blocks):
    call("avsr_multi_cast_transpose", _ptr(table_dev), n, blocks, _stream(table_dev))


def stem357_fwd(x, w, B, T, H, W):
    OH, OW = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
    y = torch.empty(B * T, OH, OW, 64, dtype=torch.bfloat16, device=x.device)
    ws = torch.empty(call("avsr_stem357_workspace_bytes") // 4 + 16, dtype=torch.float32, device=x.device)
    call("avsr_stem357_fwd", _ptr(x), _ptr(w), _ptr(y), _ptr(ws), B, T, H, W, _stream(x),
         flops=2.0 * B * T * OH * OW * 64 * 245, nbytes=_nb(x, y))
    return y


def stem357_fwd_f32s(x, w, B, T, H, W, want_stats=False):
    """Precise-mode video stem forward (f32 result, split hi / lo bf16 planes, csrc/stem.hip).  want_stats: also returns the
    per-block partial BatchNorm statistics [blocks][2][64] the kernel leaves behind (bn_finalize_parts / bn_stats_parts)."""
    OH, OW = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
    y = torch.empty(B * T, OH, OW, 64, dtype=torch.float32, device=x.device)
    ws = torch.empty(call("avsr_stem357_workspace_bytes") // 4 + 16, dtype=torch.float32, device=x.device)
    if want_stats:
        part = torch.empty(call("avsr_stem357_stat_rows", B, T, H), 2, 64, dtype=torch.float32, device=x.device)
        call("avsr_stem357_fwd_f32s_stats", _ptr(x), _ptr(w), _ptr(y), _ptr(_twin(y)), _ptr(ws), B, T, H, W, _ptr(part), part.shape[0],
             _stream(x), flops=2.0 * B * T * OH * OW * 64 * 245, nbytes=_nb(x, y))
        return y, part
    call("avsr_stem357_fwd_f32s", _ptr(x), _ptr(w), _ptr(y), _ptr(_twin(y)), _ptr(ws), B, T, H, W, _stream(x),
         flops=2.0 * B * T * OH * OW * 64 * 245, nbytes=_nb(x, y))
    return y


def stem357_wgrad(dy, x, B, T, H, W):
    dw = torch.zeros(64, 1, 5, 7, 7, dtype=torch.float32, device=x.device)
    ws = torch.empty(call("avsr_stem357_workspace_bytes") // 4 + 16, dtype=torch.float32, device=x.device)
    OH, OW = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
    call("avsr_stem357_wgrad", _ptr(dy), _ptr(x), _ptr(dw), _ptr(ws), B, T, H, W, _stream(x),
         flops=2.0 * B * T * OH * OW * 64 * 245, nbytes=_nb(dy, x))
    return dw


def gemm_bf16_tn(A, lda, B, ldb, M, N, K, C, ldc, *, accumulate=False, split_k=1, colsum_a=None):
    """C[M,N] (f32) (+)= A[K,M]^T B[K,N], bf16 operands with the contraction index as the slow dimension.
    colsum_a (f32 [M], zero-initialised): also receives the column sums of A (the bias gradient when A = dY)."""
    call("avsr_gemm_bf16_tn", _ptr(A), lda, _ptr(B), ldb, M, N, K, _ptr(C), ldc, int(accumulate), split_k,
         _ptr(zero_page(A.device)), _ptr(colsum_a), _stream(A), flops=2.0 * M * N * K,
         nbytes=2.0 * K * (M + N) + 4.0 * M * N)
    return C
