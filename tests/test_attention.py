"""Fused attention (forward, backward) against a float64 restatement of
attention.py:59-104,131-193 of the reference (mask -> finfo.min -> softmax -> zero; rel_shift as an index map)."""
import math

import pytest
import torch

from auto_avsr_amd import ops


def ref_attn(qu, qv, k, v, pos, mask, scale):
    B, T, H, D = qu.shape
    Tk = k.shape[1]
    s = torch.einsum("bihd,bjhd->bhij", qu, k)
    if pos is not None:
        g = torch.einsum("bihd,khd->bhik", qv, pos.view(-1, H, D))
        i = torch.arange(T)[:, None]
        j = torch.arange(Tk)[None, :]
        s = s + g[:, :, i, j - i + T - 1]  # == rel_shift(g)
    s = s * scale
    if mask is not None:
        mm = (mask == 0)[:, None]
        s = s.masked_fill(mm, torch.finfo(s.dtype).min)
        a = torch.softmax(s, -1).masked_fill(mm, 0.0)
    else:
        a = torch.softmax(s, -1)
    return torch.einsum("bhij,bjhd->bihd", a, v).reshape(B, T, H * D)


def make_mask(kind, B, T, Tk):
    if kind == "pad":
        lens = torch.randint(1, Tk + 1, (B,))
        lens[0] = Tk
        return (torch.arange(Tk)[None, :] < lens[:, None]).unsqueeze(1).contiguous()
    if kind == "causal":
        return torch.tril(torch.ones(T, Tk, dtype=torch.bool))[None].expand(B, -1, -1).contiguous()
    if kind == "allmasked":
        m = torch.ones(B, 1, Tk, dtype=torch.bool)
        m[0] = False
        return m
    return None


CASES = [
    # B, T, Tk, H, relpos, mask, dtype, precise, tol_fwd, tol_bwd
    (2, 70, 70, 2, True, "pad", torch.float32, True, 1e-4, 3e-4),
    (1, 130, 130, 1, True, None, torch.float32, True, 1e-4, 3e-4),
    (2, 33, 33, 2, False, "causal", torch.float32, True, 1e-4, 3e-4),
    (2, 17, 100, 2, False, "pad", torch.float32, True, 1e-4, 3e-4),
    (2, 17, 40, 2, False, "allmasked", torch.float32, True, 1e-4, 3e-4),
    (2, 70, 70, 2, True, "pad", torch.bfloat16, False, 3e-2, 0.3),
    (1, 64, 64, 1, True, None, torch.float32, False, 5e-2, 0.3),
    # bf16 forward = the transposed-formulation kernel (attention.hip AttnFwdT): every mask kind, ragged tiles, Tq != Tk
    (1, 130, 130, 1, True, None, torch.bfloat16, False, 3e-2, 0.3),
    (2, 33, 33, 2, False, "causal", torch.bfloat16, False, 3e-2, 0.3),
    (2, 17, 100, 2, False, "pad", torch.bfloat16, False, 3e-2, 0.3),
    (2, 17, 40, 2, False, "allmasked", torch.bfloat16, False, 3e-2, 0.3),
]


# the benchmarked geometry: multi-tile softmax at T = Tk = 400 with the 799-row position band, 12 heads, and the
# decoder's source attention Tq = 65 against Tk = 400; bf16 cases use relative-L2 tolerances (an absolute bound on single
# elements of a 400-term bf16 sum says little).  GPU only: the emulator needs minutes per case at this size.
BENCH_CASES = [
    # B, T, Tk, H, relpos, mask, dtype, precise, rel-L2 tol fwd, rel-L2 tol bwd
    (2, 400, 400, 3, True, "pad", torch.float32, True, 2e-5, 1e-4),
    (2, 400, 400, 3, True, "pad", torch.bfloat16, False, 1e-2, 2e-2),
    (2, 65, 400, 3, False, "pad", torch.float32, True, 2e-5, 1e-4),
    (2, 65, 400, 3, False, "pad", torch.bfloat16, False, 1e-2, 2e-2),
    (2, 65, 65, 3, False, "causal", torch.bfloat16, False, 1e-2, 2e-2),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", BENCH_CASES, ids=[f"b{i}" for i in range(len(BENCH_CASES))])
def test_attention_bench_geometry(case):
    B, T, Tk, H, relpos, mkind, dtype, precise, tol_f, tol_b = case
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 1000 + T + Tk)
    D = 64
    qu, qv = torch.randn(B, T, H, D).to(dtype), torch.randn(B, T, H, D).to(dtype)
    k, v = torch.randn(B, Tk, H, D).to(dtype), torch.randn(B, Tk, H, D).to(dtype)
    pos = torch.randn(2 * T - 1, H * D).to(dtype) if relpos else None
    mask = make_mask(mkind, B, T, Tk)
    dout = torch.randn(B, T, H * D).to(dtype)
    scale = 1 / math.sqrt(D)
    d = lambda t: None if t is None else t.to(dev)
    out, lse = ops.attention_fwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), scale, precise=precise)
    leaf = lambda t: None if t is None else t.double().requires_grad_()
    rqu, rqv, rk, rv, rpos = leaf(qu), leaf(qv), leaf(k), leaf(v), leaf(pos)
    ref = ref_attn(rqu, rqv, rk, rv, rpos, mask, scale)
    rel = lambda a, b: float((a.cpu().double() - b.detach()).norm() / b.detach().norm())
    assert rel(out, ref) < tol_f
    ref.backward(dout.double())
    dqu, dqv, dk, dv, dpos = ops.attention_bwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), out,
                                               lse, d(dout), scale, precise=precise)
    errs = {"dqu": rel(dqu, rqu.grad), "dk": rel(dk, rk.grad), "dv": rel(dv, rv.grad)}
    if relpos:
        errs.update(dqv=rel(dqv, rqv.grad), dpos=rel(dpos, rpos.grad))
    assert max(errs.values()) < tol_b, errs


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_attention_fwd_bwd(dev, case):
    B, T, Tk, H, relpos, mkind, dtype, precise, tol_f, tol_b = case
    torch.manual_seed(B * 1000 + T)
    D = 64
    qu, qv = torch.randn(B, T, H, D).to(dtype), torch.randn(B, T, H, D).to(dtype)
    k, v = torch.randn(B, Tk, H, D).to(dtype), torch.randn(B, Tk, H, D).to(dtype)
    pos = torch.randn(2 * T - 1, H * D).to(dtype) if relpos else None
    mask = make_mask(mkind, B, T, Tk)
    dout = torch.randn(B, T, H * D).to(dtype)
    scale = 1 / math.sqrt(D)
    d = lambda t: None if t is None else t.to(dev)
    out, lse = ops.attention_fwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), scale, precise=precise)
    leaf = lambda t: None if t is None else t.double().requires_grad_()
    rqu, rqv, rk, rv, rpos = leaf(qu), leaf(qv), leaf(k), leaf(v), leaf(pos)
    ref = ref_attn(rqu, rqv, rk, rv, rpos, mask, scale)
    assert (out.cpu().double() - ref).abs().max() < tol_f
    ref.backward(dout.double())
    dqu, dqv, dk, dv, dpos = ops.attention_bwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), out,
                                               lse, d(dout), scale, precise=precise)
    assert (dqu.cpu().double() - rqu.grad).abs().max() < tol_b
    assert (dk.cpu().double() - rk.grad).abs().max() < tol_b
    assert (dv.cpu().double() - rv.grad).abs().max() < tol_b
    if relpos:
        assert (dqv.cpu().double() - rqv.grad).abs().max() < tol_b
        assert (dpos.cpu().double() - rpos.grad).abs().max() < tol_b


@pytest.mark.parametrize("dtype,precise,T", [(torch.float32, True, 70), (torch.bfloat16, False, 130)])
def test_attention_bwd_emits_query_and_bias_gradients(dev, dtype, precise, T):
    """dq_sum / du / dv of avsr_attention_bwd_dq: the gradient of q (= dqu + dqv) written straight into a pitched view
    (the q third of a fused d(qkv) buffer) and the pos_bias_u / pos_bias_v gradients (column sums of dqu / dqv over
    (batch, time), attention.py:171-177 under autograd) -- against float64 autograd; ragged last query tile."""
    torch.manual_seed(T)
    B, H, D = 2, 3, 64
    q = torch.randn(B, T, H, D).to(dtype)
    u, v_b = torch.randn(H, D), torch.randn(H, D)
    qu, qv = (q.float() + u).to(dtype), (q.float() + v_b).to(dtype)
    k, v = torch.randn(B, T, H, D).to(dtype), torch.randn(B, T, H, D).to(dtype)
    pos = torch.randn(2 * T - 1, H * D).to(dtype)
    mask = make_mask("pad", B, T, T)
    dout = torch.randn(B, T, H * D).to(dtype)
    scale = 1 / math.sqrt(D)
    d = lambda t: t.to(dev)
    out, lse = ops.attention_fwd(d(qu), d(qv), d(k), d(v), d(pos), d(mask), scale, precise=precise)
    rqu, rqv = qu.double().requires_grad_(), qv.double().requires_grad_()
    ref = ref_attn(rqu, rqv, k.double(), v.double(), pos.double(), mask, scale)
    ref.backward(dout.double())
    dqkv = torch.full((B, T, 3, H, D), 7.0, dtype=dtype, device=dev)  # k / v thirds must stay untouched
    du = torch.zeros(H * D, dtype=torch.float32, device=dev)
    dv = torch.zeros(H * D, dtype=torch.float32, device=dev)
    dqu, dqv, pd, ds = ops.attention_bwd_dq(d(qu), d(qv), d(k), d(v), d(pos), d(mask), out, lse, d(dout), scale,
                                            precise=precise, dq_sum=dqkv[:, :, 0], du=du, dv=dv)
    assert dqu is None and dqv is None
    rel = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    tol = 1e-4 if precise else 2e-2
    assert rel(dqkv[:, :, 0], rqu.grad + rqv.grad) < tol
    assert rel(du.view(H, D), rqu.grad.sum((0, 1))) < tol and rel(dv.view(H, D), rqv.grad.sum((0, 1))) < tol
    assert bool((dqkv[:, :, 1:] == 7.0).all())
    # same launch without the fused outputs: identical pd / ds, and dqu + dqv equals dq_sum to rounding
    dqu2, dqv2, pd2, ds2 = ops.attention_bwd_dq(d(qu), d(qv), d(k), d(v), d(pos), d(mask), out, lse, d(dout), scale,
                                                precise=precise)
    assert torch.equal(pd[..., :T], pd2[..., :T]) and torch.equal(ds[..., :T], ds2[..., :T])  # (pad columns are never read)
    assert rel(dqkv[:, :, 0].float(), (dqu2.float() + dqv2.float()).cpu().double()) < (1e-6 if precise else 8e-3)


@pytest.mark.parametrize("relpos", [False, True])
def test_attention_bf16_forward_kernels_agree(dev, relpos):
    """The bf16 forward has two kernels (transposed formulation = default; generic = avsr_tune knob 8): same output to bf16
    rounding, same log-sum-exp, and -- with dropout on -- the SAME keep mask (the backward kernel recomputes it from the
    (row, key) index, so both forward kernels must draw it identically): compared through the dropped positions of a
    V = identity probe."""
    torch.manual_seed(11)
    B, T, H, D = 2, 100, 2, 64
    qu, qv, k = (torch.randn(B, T, H, D).bfloat16() for _ in range(3))
    v = torch.randn(B, T, H, D).bfloat16()
    pos = torch.randn(2 * T - 1, H * D).bfloat16() if relpos else None
    mask = make_mask("pad", B, T, T)
    d = lambda t: None if t is None else t.to(dev)
    res = []
    try:
        for knob in (0, 1):
            ops.tune(8, knob)
            o, l = ops.attention_fwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), 0.125)
            vI = torch.zeros(B, T, H, D)
            vI[:, :64, :, :] = torch.eye(64).view(1, 64, 1, 64)
            od, _ = ops.attention_fwd(d(qu), d(qv) if relpos else None, d(k), d(vI.bfloat16()), d(pos), d(mask), 0.125,
                                      drop_p=0.25, seed=99)
            res.append((o.float().cpu(), l.cpu(), od.float().cpu()))
    finally:
        ops.tune(8, 0)
    (o0, l0, d0), (o1, l1, d1) = res
    assert (o0 - o1).abs().max() < 2e-2 * max(1.0, float(o1.abs().max()))
    assert (l0 - l1).abs().max() < 1e-3
    assert torch.equal(d0 == 0, d1 == 0), "the two forward kernels drew different dropout masks"
    assert 0.15 < float((d0 == 0).float().mean()) < 0.6


@pytest.mark.parametrize("relpos,mkind,T,Tk", [(True, "pad", 100, 100), (False, "causal", 70, 70), (False, "pad", 33, 130),
                                                (True, None, 64, 64)])
def test_attention_bf16_backward_kernels_agree(dev, relpos, mkind, T, Tk):
    """The bf16 backward-dq pass has two kernels (transposed formulation = default; generic = avsr_tune knob 9): same dqu /
    dqv, same pd = dropout(P) and ds = scale * dS (what the key / value side contracts), same fused dq_sum / du / dv, with
    dropout on (identical keep mask) -- every mask kind, ragged tiles, Tq != Tk."""
    torch.manual_seed(T + Tk)
    B, H, D = 2, 2, 64
    qu, qv = torch.randn(B, T, H, D).bfloat16(), torch.randn(B, T, H, D).bfloat16()
    k, v = torch.randn(B, Tk, H, D).bfloat16(), torch.randn(B, Tk, H, D).bfloat16()
    pos = torch.randn(2 * T - 1, H * D).bfloat16() if relpos else None
    mask = make_mask(mkind, B, T, Tk)
    dout = torch.randn(B, T, H * D).bfloat16()
    d = lambda t: None if t is None else t.to(dev)
    out, lse = ops.attention_fwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), 0.125, drop_p=0.2, seed=5)
    res = []
    try:
        for knob in (0, 1):
            ops.tune(9, knob)
            dqu, dqv, pd, ds = ops.attention_bwd_dq(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), out, lse,
                                                    d(dout), 0.125, drop_p=0.2, seed=5)
            extra = None
            if relpos:
                dq = torch.zeros(B, T, H, D, dtype=torch.bfloat16, device=dev)
                du, dv = torch.zeros(H * D, device=dev), torch.zeros(H * D, device=dev)
                ops.attention_bwd_dq(d(qu), d(qv), d(k), d(v), d(pos), d(mask), out, lse, d(dout), 0.125, drop_p=0.2, seed=5,
                                     dq_sum=dq, du=du, dv=dv)
                extra = (dq.float().cpu(), du.cpu(), dv.cpu())
            res.append((dqu.float().cpu(), None if dqv is None else dqv.float().cpu(), pd[..., :Tk].float().cpu(),
                        ds[..., :Tk].float().cpu(), extra))
    finally:
        ops.tune(9, 0)
    a, g = res
    rel = lambda x, y: float((x - y).norm() / (y.norm() + 1e-30))
    assert torch.equal(a[2] == 0, g[2] == 0), "different dropout / mask pattern in pd"
    assert rel(a[2], g[2]) < 4e-3 and rel(a[3], g[3]) < 8e-3, (rel(a[2], g[2]), rel(a[3], g[3]))
    assert rel(a[0], g[0]) < 8e-3, rel(a[0], g[0])
    if relpos:
        assert rel(a[1], g[1]) < 8e-3, rel(a[1], g[1])
        for x, y in zip(a[4], g[4]):
            assert rel(x, y) < 8e-3, rel(x, y)


@pytest.mark.parametrize("relpos,B,T,Tk,H", [(True, 3, 100, 100, 2), (True, 1, 64, 64, 1), (False, 2, 33, 130, 2),
                                             (False, 2, 70, 70, 3), (True, 2, 129, 129, 1)])
def test_attention_bwd_kv_fast_matches_generic(dev, relpos, B, T, Tk, H):
    """Key / value side of the bf16 backward: the k-major tile kernel (attention_kv.hip, the default; dpos contracted over
    (batch, query) jointly, skewed dS rows read at 2-byte-aligned addresses) vs the generic batched TN path (avsr_tune knob
    10 = 1) on the SAME stored pd / ds: dK / dV to bf16 rounding, dpos (f32) to summation order; dpos lands in a pitched
    column block and ACCUMULATES (a second call doubles it); ragged tiles, Tq != Tk, batch sum.  Runs on the emulator and,
    since round 3, on the MI355X (its first hardware runs faulted: see the history note in attention_kv.hip)."""
    _kv_fast_vs_generic(dev, relpos, B, T, Tk, H)


@pytest.mark.gpu
@pytest.mark.parametrize("relpos,B,T,Tk,H", [(True, 4, 400, 400, 12), (False, 4, 65, 400, 12), (False, 4, 65, 65, 12)])
def test_attention_bwd_kv_fast_bench_geometry(relpos, B, T, Tk, H):
    """The same comparison at the benchmark's geometries (encoder T = 400, decoder source / self attention), MI355X only."""
    from auto_avsr_amd import _lib

    _lib._lib = None
    assert not _lib.lib().is_emulator
    _kv_fast_vs_generic(torch.device("cuda:0"), relpos, B, T, Tk, H)


def _kv_fast_vs_generic(dev, relpos, B, T, Tk, H):
    torch.manual_seed(B * 100 + T + Tk)
    D = 64
    qu, qv = torch.randn(B, T, H, D).bfloat16(), torch.randn(B, T, H, D).bfloat16()
    k, v = torch.randn(B, Tk, H, D).bfloat16(), torch.randn(B, Tk, H, D).bfloat16()
    pos = torch.randn(2 * T - 1, H * D).bfloat16() if relpos else None
    mask = make_mask("pad", B, T, Tk)
    dout = torch.randn(B, T, H * D).bfloat16()
    d = lambda t: None if t is None else t.to(dev)
    out, lse = ops.attention_fwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), 0.125)
    res = []
    try:
        for knob in (0, 1):  # 0 = the tile kernel (default), 1 = generic
            ops.tune(10, knob)
            wide = torch.zeros(2 * T - 1, 3 * H * D, device=dev) if relpos else None
            kw = dict(dpos_out=wide[:, H * D:2 * H * D]) if relpos else {}
            _, _, dk, dv, dpos = ops.attention_bwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), out, lse,
                                                   d(dout), 0.125, **kw)
            if relpos:
                once = dpos.clone()
                ops.attention_bwd(d(qu), d(qv), d(k), d(v), d(pos), d(mask), out, lse, d(dout), 0.125, **kw)
                assert float((wide[:, H * D:2 * H * D] - 2 * once).abs().max()) <= 1e-5 * max(1.0, float(once.abs().max()))
                assert float(wide[:, :H * D].abs().max()) == 0.0 and float(wide[:, 2 * H * D:].abs().max()) == 0.0
                dpos = once
            res.append((dk.float().cpu(), dv.float().cpu(), None if dpos is None else dpos.cpu()))
    finally:
        ops.tune(10, 0)
    (dk0, dv0, dp0), (dk1, dv1, dp1) = res
    rel = lambda x, y: float((x - y).norm() / (y.norm() + 1e-30))
    assert rel(dk0, dk1) < 5e-3 and rel(dv0, dv1) < 5e-3, (rel(dk0, dk1), rel(dv0, dv1))
    if relpos:
        assert rel(dp0, dp1) < 1e-5, rel(dp0, dp1)


@pytest.mark.parametrize("relpos,mkind,T,Tk,dtype", [(False, "pad", 70, 300, torch.bfloat16), (False, "pad", 33, 257, torch.float16),
                                                   (True, "pad", 130, 130, torch.bfloat16), (False, "causal", 200, 200, torch.float16)])
def test_attention_fwd_key_split(dev, relpos, mkind, T, Tk, dtype):
    """Round 6: the forward kernel with the key range split over two wave groups (source attention of the decoder: few blocks, long
    key chains) against the unsplit kernel and float64 math: same output and log-sum-exp up to the rounding of the merge -- ragged
    halves, a group with no tile at all, padding masks, rows past Tq; avsr_tune knob 8 = 3 forces the split, 2 forbids it."""
    B, H, D = 2, 2, 64
    torch.manual_seed(T + Tk)
    qu, qv = torch.randn(B, T, H, D).to(dtype), torch.randn(B, T, H, D).to(dtype)
    k, v = torch.randn(B, Tk, H, D).to(dtype), torch.randn(B, Tk, H, D).to(dtype)
    pos = torch.randn(2 * T - 1, H * D).to(dtype) if relpos else None
    mask = make_mask(mkind, B, T, Tk)
    scale = 1 / math.sqrt(D)
    d = lambda t: None if t is None else t.to(dev)
    res = {}
    try:
        for knob in (2, 3):
            ops.tune(8, knob)
            out, lse = ops.attention_fwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), scale)
            res[knob] = (out.float().cpu(), lse.float().cpu())
    finally:
        ops.tune(8, 0)
    ref = ref_attn(qu.double(), qv.double() if relpos else None, k.double(), v.double(), None if pos is None else pos.double(), mask, scale)
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    assert rel(res[3][0], ref) < 1e-2 and rel(res[2][0], ref) < 1e-2
    assert rel(res[3][0], res[2][0].double()) < 4e-3                  # one 16-bit rounding of the output apart
    assert (res[3][1] - res[2][1]).abs().max() < 1e-4                  # the log-sum-exp the backward pass reads


def test_attention_dropout_consistency(dev):
    """Dropout on the probabilities: forward and backward must draw the same keep-mask (finite-difference free
    check: with V = I-like probes the output equals the dropped probabilities that the backward re-creates)."""
    torch.manual_seed(5)
    B, T, H, D = 1, 64, 1, 64
    qu, k = torch.randn(B, T, H, D), torch.randn(B, T, H, D)
    v = torch.zeros(B, T, H, D)
    v[0, :, 0, :] = torch.eye(T, D)
    scale = 0.125
    d = lambda t: t.to(dev)
    out, lse = ops.attention_fwd(d(qu), None, d(k), d(v), None, None, scale, precise=True, drop_p=0.3, seed=1234)
    dout = torch.zeros(B, T, H * D)
    _, _, pd, _ = ops.attention_bwd_dq(d(qu), None, d(k), d(v), None, None, out, lse, d(dout), scale, precise=True,
                                       drop_p=0.3, seed=1234)
    pd = pd.cpu()[0, 0, :, :T]
    assert (pd - out.cpu()[0]).abs().max() < 1e-4
    frac_dropped = (pd == 0).float().mean().item()
    assert 0.2 < frac_dropped < 0.4
    p_ref = torch.softmax(torch.einsum("id,jd->ij", qu[0, :, 0].double(), k[0, :, 0].double()) * scale, -1)
    kept = pd != 0
    assert ((pd.double() - p_ref / 0.7)[kept]).abs().max() < 1e-4
