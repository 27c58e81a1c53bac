"""The "mixed" numerical mode (functional.set_mode("mixed")): forward kernels on IEEE-half (f16) operands with bf16 twins for the
backward pass, and the per-component forward policy.

Kernel level: every f16 forward entry point (include/avsr_hip.h, section "mixed") against float64 math on the SAME f16-rounded
inputs -- exact products, f32 accumulation -- and its bf16 twin against the bf16 rounding of the f16-path result.
Module level: a small E2E instance against the fp32 oracle (oracle/avsr_oracle.py) for several policies.
Each test runs on the host emulator build (CPU suite) and on the gfx950 build (-m gpu)."""
import math
import os
import sys

import pytest
import torch

from auto_avsr_amd import functional as AF
from auto_avsr_amd import ops

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)


def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture
def twins(monkeypatch):
    """Switch the producer-side twins on outside an autograd function (what functional.set_mode('mixed') does in a step)."""
    made = []

    def make(y):
        t = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
        made.append((y, t))
        return t

    monkeypatch.setattr(ops, "TWIN", make)
    return made


@pytest.mark.parametrize("tile", [0, 1, 4, 7])
@pytest.mark.parametrize("shape", [(150, 70, 192), (257, 300, 448), (130, 136, 64)])
def test_gemm_h16_nt(dev, tile, shape, twins):
    M, N, K = shape
    torch.manual_seed(M + tile)
    A, B = torch.randn(M, K).half(), (0.1 * torch.randn(N, K)).half()
    ref = A.double() @ B.double().t()
    C = torch.zeros(M, N + 3, device=dev)
    ops.gemm_h16_nt(A.to(dev), K, B.to(dev), K, M, N, K, C, N + 3, tile=tile)
    assert ((C.cpu()[:, :N].double() - ref).abs().max() / ref.abs().max()) < 2e-6  # exact products, f32 accumulation
    assert C.cpu()[:, N:].abs().max() == 0
    # epilogue (bias, relu, alpha, f32 residual) + f16 output + its bf16 twin
    bias, resid = torch.randn(N), torch.randn(M, N)
    C2 = torch.zeros(M, N, device=dev, dtype=torch.float16)
    ops.gemm_h16_nt(A.to(dev), K, B.to(dev), K, M, N, K, C2, N, bias=bias.to(dev), act=1, alpha=0.5, resid=resid.to(dev), ldr=N,
                    tile=tile, twin=(N % 8 == 0))
    ref2 = torch.relu(ref + bias.double()) * 0.5 + resid.double()
    assert ((C2.cpu().double() - ref2).abs().max() / ref2.abs().max()) < 1.5e-3  # one f16 rounding (2^-11)
    if N % 8 == 0:
        (y, tw), = twins
        assert y is C2 and rel(tw.float(), ref2) < 4e-3 and (tw.float().cpu() - ref2.float()).abs().max() < 1.2e-2 * ref2.abs().max()
    # an 11-bit significand is the point: the same contraction on bf16 roundings of the SAME f32 data is ~8x further off
    Af, Bf = torch.randn(M, K), 0.1 * torch.randn(N, K)
    exact = Af.double() @ Bf.double().t()
    Ch = torch.zeros(M, N, device=dev)
    ops.gemm_h16_nt(Af.half().to(dev), K, Bf.half().to(dev), K, M, N, K, Ch, N, tile=tile)
    Cb = torch.zeros(M, N, device=dev)
    ops.gemm_bf16_nt(Af.bfloat16().to(dev), K, Bf.bfloat16().to(dev), K, M, N, K, Cb, N)
    assert rel(Ch, exact) < 6e-4 and rel(Cb, exact) > 4 * rel(Ch, exact)


def test_layernorm_headbias_casts_f16(dev, twins):
    torch.manual_seed(0)
    rows, cols = 37, 768
    x, g, b = torch.randn(rows, cols), torch.randn(cols), torch.randn(cols)
    ref = torch.nn.functional.layer_norm(x, (cols,), g, b, 1e-12)
    y, mean, rstd = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), torch.float16, twin=True)
    assert y.dtype == torch.float16 and (y.float().cpu() - ref).abs().max() < 4e-3
    (yy, tw), = twins
    assert yy is y and (tw.float().cpu() - ref).abs().max() < 3e-2
    y32, mean32, rstd32 = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), torch.float32)
    assert torch.equal(mean, mean32) and torch.equal(rstd, rstd32)
    twins.clear()
    # q + pos_bias_u / q + pos_bias_v on an f16 slice of a pitched buffer
    q = torch.randn(rows, 3 * cols).half()
    u, v = torch.randn(cols), torch.randn(cols)
    o1, o2 = ops.head_bias_fwd(q.to(dev), 3 * cols, rows, cols, u.to(dev), v.to(dev))
    assert o1.dtype == torch.float16
    assert (o1.float().cpu() - (q[:, :cols].float() + u)).abs().max() < 4e-3
    assert (o2.float().cpu() - (q[:, :cols].float() + v)).abs().max() < 4e-3
    assert len(twins) == 2 and (twins[0][1].float().cpu() - (q[:, :cols].float() + u)).abs().max() < 3e-2
    # casts: f32 -> f16 (round to nearest even, saturating), f16 -> f32 exact, f16 -> bf16
    z = torch.cat([torch.randn(1000), torch.tensor([1e6, -1e6, 65504.0, 6e-8, 0.0, 1.0 + 2 ** -11, 1.0 + 3 * 2 ** -11])])
    z = torch.cat([z, torch.zeros((-z.numel()) % 8)])
    h = ops.scale_dropout(z.to(dev), torch.float16)
    want = z.clamp(-65504.0, 65504.0).half()
    assert torch.equal(h.cpu(), want)
    assert torch.equal(ops.scale_dropout(h, torch.float32).cpu(), want.float())
    assert torch.equal(ops.scale_dropout(h, torch.bfloat16).cpu(), want.float().bfloat16())


@pytest.mark.parametrize("case", [(2, 70, 70, 2, True, "pad"), (1, 130, 130, 1, True, None), (2, 33, 33, 2, False, "causal"),
                                  (2, 17, 100, 2, False, "pad"), (2, 17, 40, 2, False, "allmasked")])
def test_attention_fwd_f16(dev, case, twins):
    """f16 forward on the transposed-formulation kernel: every mask kind, ragged tiles, Tq != Tk; against float64 math on the same
    f16 inputs.  11-bit operands: an order of magnitude inside the bf16 kernel's tolerance (3e-2 in test_attention.py)."""
    from test_attention import make_mask, ref_attn

    B, T, Tk, H, relpos, mkind = case
    torch.manual_seed(5)
    qu, qv = torch.randn(B, T, H, 64).half(), torch.randn(B, T, H, 64).half()
    k, v = torch.randn(B, Tk, H, 64).half(), torch.randn(B, Tk, H, 64).half()
    pos = torch.randn(2 * T - 1, H * 64).half() if relpos else None
    mask = make_mask(mkind, B, T, Tk)
    d = lambda t: None if t is None else t.to(dev)
    ref = ref_attn(qu.double(), qv.double(), k.double(), v.double(), None if pos is None else pos.double(), mask, 0.125)
    out, lse = ops.attention_fwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), 0.125)
    assert out.dtype == torch.float16
    assert (out.float().cpu().double() - ref).abs().max() < 4e-3 * max(1.0, float(ref.abs().max()))
    (o, tw), = twins
    assert o is out and (tw.float().cpu().double() - ref).abs().max() < 3e-2 * max(1.0, float(ref.abs().max()))
    # the log-sum-exp the bf16 backward kernel pairs with: equal to the f32 kernel's on the same (f16-exact) inputs
    out32, lse32 = ops.attention_fwd(d(qu.float()), d(qv.float()) if relpos else None, d(k.float()), d(v.float()),
                                     d(None if pos is None else pos.float()), d(mask), 0.125, precise=True)
    assert (lse.cpu() - lse32.cpu()).abs().max() < 2e-3


@pytest.mark.gpu
def test_attention_fwd_f16_bench_geometry(twins):
    from test_attention import make_mask, ref_attn

    dev = torch.device("cuda")
    for (B, T, Tk, H, relpos, mkind) in [(2, 400, 400, 3, True, "pad"), (2, 65, 400, 3, False, "pad"), (2, 65, 65, 3, False, "causal")]:
        torch.manual_seed(6)
        qu, qv = torch.randn(B, T, H, 64).half(), torch.randn(B, T, H, 64).half()
        k, v = torch.randn(B, Tk, H, 64).half(), torch.randn(B, Tk, H, 64).half()
        pos = torch.randn(2 * T - 1, H * 64).half() if relpos else None
        mask = make_mask(mkind, B, T, Tk)
        d = lambda t: None if t is None else t.to(dev)
        ref = ref_attn(qu.double(), qv.double(), k.double(), v.double(), None if pos is None else pos.double(), mask, 0.125)
        out, _ = ops.attention_fwd(d(qu), d(qv) if relpos else None, d(k), d(v), d(pos), d(mask), 0.125)
        assert rel(out.float(), ref) < 1.5e-3  # (the bf16 kernel: 1e-2)


def test_dwconv_bn_f16(dev, twins):
    """GLU + depthwise conv and the single-launch BatchNorm1d + Swish of the convolution module on f16 activations."""
    torch.manual_seed(2)
    B, T, C, K = 2, 50, 128, 31
    a = torch.randn(B * T, 2 * C).half()
    w, bias = 0.2 * torch.randn(C, K), torch.randn(C)
    glu = a[:, :C].double() * torch.sigmoid(a[:, C:].double())
    ref = torch.nn.functional.conv1d(glu.view(B, T, C).transpose(1, 2), w.double().unsqueeze(1), bias.double(), padding=15,
                                     groups=C).transpose(1, 2)
    y = ops.dwconv(a.to(dev), w.to(dev), bias.to(dev), B, T, C, K, glu_in=True)
    assert y.dtype == torch.float16 and rel(y.float(), ref) < 6e-4
    assert rel(twins[0][1].float(), ref) < 4e-3
    twins.clear()
    g, b = torch.rand(C) + 0.5, torch.randn(C)
    rm, rv, nbt = torch.zeros(C), torch.ones(C), torch.zeros((), dtype=torch.int64)
    x = y.cpu().view(B * T, C)
    bn = torch.nn.BatchNorm1d(C).double()
    bn.weight.data, bn.bias.data = g.double(), b.double()
    refy = torch.nn.functional.silu(bn(x.double()))
    s, mean, invstd = ops.bn_small_fwd(y, B * T, C, g.to(dev), b.to(dev), 1e-5, 0.1, rm.to(dev), rv.to(dev), nbt.to(dev), 1)
    assert s.dtype == torch.float16 and rel(s.float(), refy) < 6e-4 and rel(twins[0][1].float(), refy) < 4e-3
    assert (mean.cpu().double() - x.double().mean(0)).abs().max() < 1e-5
    # the multi-launch path (what cross-rank BatchNorm runs): statistics + apply on the f16 tensor
    twins.clear()
    m2, i2 = ops.bn_stats_finalize(y, B * T, C, 1e-5, 0.1, rm.to(dev), rv.to(dev), None)
    s2 = ops.bn_act_fwd(y, None, m2, i2, g.to(dev), b.to(dev), B * T, C, 1)
    assert rel(s2.float(), refy) < 6e-4 and rel(twins[0][1].float(), refy) < 4e-3
    flat = ops.bn_stats(y, B * T, C, with_count=True)
    assert float(flat[-1]) == B * T


def _planes(w):
    """hi / scaled-lo f16 planes of an f32 weight as the kernels define them (csrc/prims.h f2h_lo)."""
    hi = w.half()
    return hi, ((w - hi.float()) * 2048.0).half()


@pytest.mark.parametrize("tile", [0, 1, 21, 7, 2, 4, 5, 8])
@pytest.mark.parametrize("shape", [(150, 70, 192), (257, 300, 448), (130, 136, 64)])
def test_gemm_h16x2_nt(dev, tile, shape, twins):
    """Two weight planes (round 5): products with the EXACT f32 weight up to ~2^-22, f16 activations.  Against float64 math on
    the f16-rounded activations and the unrounded weight; the one-plane kernel on the same data is >= 20x further off.  The planes
    come from the multi-tensor cast the training step uses (functional._w_h16) and, for the pitch / slice handling, from torch."""
    M, N, K = shape
    torch.manual_seed(M + tile)
    A, W = torch.randn(M, K).half(), 0.1 * torch.randn(N, K)
    W[0, :8] = torch.tensor([0.0, 1e-7, -3e-6, 6.1e-5, 0.25, -0.2500001, 1.0, 3.7e-3])  # zeros, sub-f16-normal values, binade edges
    ref = A.double() @ W.double().t()
    AF.invalidate_weight_cache()
    Wd = W.to(dev)
    buf = AF._w_h16(Wd)
    hi, lo = _planes(W)
    assert buf.shape == (N, 2, K) and torch.equal(buf[:, 0].cpu(), hi) and torch.equal(buf[:, 1].cpu(), lo)
    C = torch.zeros(M, N + 3, device=dev)
    ops.gemm_h16_nt(A.to(dev), K, buf[:, 0], 2 * K, M, N, K, C, N + 3, tile=tile, B_lo=buf[:, 1])
    e2 = float((C.cpu()[:, :N].double() - ref).abs().max() / ref.abs().max())
    assert e2 < 3e-6 and C.cpu()[:, N:].abs().max() == 0, e2
    C1 = torch.zeros(M, N, device=dev)
    ops.gemm_h16_nt(A.to(dev), K, buf[:, 0], 2 * K, M, N, K, C1, N, tile=1 if tile == 21 else tile)
    assert rel(C1, ref) > 20 * rel(C[:, :N], ref)
    # epilogue + f16 output + twin, through the host helper the sub-layers call
    bias, resid = torch.randn(N), torch.randn(M, N)
    C2 = torch.zeros(M, N, device=dev, dtype=torch.float16)
    AF._h16_nt(A.to(dev), K, buf, M, N, K, C2, N, planes=2, bias=bias.to(dev), act=1, alpha=0.5, resid=resid.to(dev), ldr=N, tile=tile,
               twin=(N % 8 == 0))
    ref2 = torch.relu(ref + bias.double()) * 0.5 + resid.double()
    assert ((C2.cpu().double() - ref2).abs().max() / ref2.abs().max()) < 1.5e-3  # one f16 rounding of the result
    if N % 8 == 0:
        (y, tw), = twins
        assert y is C2 and rel(tw.float(), ref2) < 4e-3
    AF.invalidate_weight_cache()


@pytest.mark.parametrize("geom", [(3, 11, 11, 64, 128, 3, 2), (2, 6, 6, 128, 128, 3, 1), (3, 11, 11, 64, 128, 1, 2), (2, 22, 22, 64, 64, 3, 1)])
@pytest.mark.parametrize("tile", [0, 4, 5])
def test_conv2d_h16x2(dev, geom, tile, twins):
    """Implicit-GEMM convolution with two filter planes: exact filter, f16 activations (float64 math on the same data)."""
    N, H, W, Cin, Cout, KH, stride = geom
    pad = (KH - 1) // 2
    torch.manual_seed(N + H)
    x = torch.randn(N, H, W, Cin).half()
    w = (0.1 * torch.randn(Cout, Cin, KH, KH))
    AF.invalidate_weight_cache()
    wd = w.to(dev)
    buf = AF._w_conv_h16(wd)
    wperm = w.permute(0, 2, 3, 1).reshape(Cout, -1)
    hi, lo = _planes(wperm)
    assert buf.shape == (Cout, 2, KH * KH * Cin) and torch.equal(buf[:, 0].cpu(), hi) and torch.equal(buf[:, 1].cpu(), lo)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    ops.tune(18, tile)
    try:
        y = ops.conv2d_fwd(x.to(dev), buf, N, H, W, Cin, Cout, KH, KH, stride, pad, pad, False, wp_planes=2)
    finally:
        ops.tune(18, 0)
    assert y.dtype == torch.float16 and rel(y.float(), ref) < 2.5e-4  # (the f16 rounding of the OUTPUT: 2^-12 rms)
    (yy, tw), = twins
    assert yy is y and rel(tw.float(), ref) < 4e-3
    y1 = ops.conv2d_fwd(x.to(dev), buf, N, H, W, Cin, Cout, KH, KH, stride, pad, pad, False, wp_planes=1)
    ref1 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.half().double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    assert rel(y1.float(), ref1) < 2.5e-4  # one plane = the hi rows of the same image
    AF.invalidate_weight_cache()


@pytest.mark.parametrize("geom", [(3, 11, 11, 64, 128, 3, 2), (2, 6, 6, 128, 128, 3, 1), (3, 11, 11, 64, 128, 1, 2), (2, 22, 22, 64, 64, 3, 1)])
def test_conv2d_h16(dev, geom, twins):
    """Implicit-GEMM convolution on f16 operands (trunk stages of the mixed mode) vs torch on the same f16-rounded data."""
    N, H, W, Cin, Cout, KH, stride = geom
    pad = (KH - 1) // 2
    torch.manual_seed(N + H)
    x = torch.randn(N, H, W, Cin).half()
    w = (0.1 * torch.randn(Cout, Cin, KH, KH))
    wp = ops.conv_weight_permute(w.to(dev), torch.float16)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.half().double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    y = ops.conv2d_fwd(x.to(dev), wp, N, H, W, Cin, Cout, KH, KH, stride, pad, pad, False)
    assert y.dtype == torch.float16 and rel(y.float(), ref) < 5e-4
    (yy, tw), = twins
    assert yy is y and rel(tw.float(), ref) < 4e-3


@pytest.mark.parametrize("geom", [(3, 11, 11, 64, 128, 3, 2), (5, 22, 22, 64, 64, 3, 1), (3, 11, 11, 64, 128, 1, 2)])
def test_conv_epilogue_leaves_batchnorm_statistics(dev, geom):
    """avsr_conv2d_f32s_stats: the split-plane convolution's epilogue leaves per-column sums / sums of squares of every 128-row
    tile of its output; avsr_bn_finalize_parts turns them into (mean, invstd) + running statistics, avsr_bn_stats_parts into the
    cross-rank payload -- against the stand-alone statistics pass over the same output and torch's batch_norm."""
    N, H, W, Cin, Cout, KH, stride = geom
    pad = (KH - 1) // 2
    torch.manual_seed(N * H)
    x = (torch.randn(N, H, W, Cin) + 0.3).to(dev)
    w = (0.1 * torch.randn(Cout, Cin, KH, KH)).to(dev)
    wp = ops.conv_weight_permute(w, torch.float32)
    OH = (H + 2 * pad - KH) // stride + 1
    part = torch.full((ops.bn_stat_tiles(N * OH * OH), 2, Cout), float("nan"), device=dev)  # (every row must be written)
    assert ops.conv2d_takes_stats(x, wp, Cin, KH, KH, True)
    y = ops.conv2d_fwd(x, wp, N, H, W, Cin, Cout, KH, KH, stride, pad, pad, True, stats=part)
    y0 = ops.conv2d_fwd(x, wp, N, H, W, Cin, Cout, KH, KH, stride, pad, pad, True)
    assert torch.equal(y, y0)
    rows = y.numel() // Cout
    y2 = y.view(rows, Cout).double()
    assert rel(part[:, 0].sum(0).double(), y2.sum(0)) < 1e-5 and rel(part[:, 1].sum(0).double(), (y2 * y2).sum(0)) < 1e-5
    rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    rm0, rv0 = rm.clone(), rv.clone()
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    mean, invstd = ops.bn_finalize_parts(part, rows, Cout, 1e-5, 0.1, rm, rv, nbt)
    mean0, invstd0 = ops.bn_stats_finalize(y.view(rows, Cout), rows, Cout, 1e-5, 0.1, rm0, rv0)
    assert rel(mean, mean0) < 1e-5 and rel(invstd, invstd0) < 1e-5 and rel(rm, rm0) < 1e-5 and rel(rv, rv0) < 1e-5 and int(nbt) == 1
    flat = ops.bn_stats_parts(part, rows, Cout)
    assert float(flat[3 * Cout]) == rows and float(flat[:Cout].abs().max()) == 0.0
    assert rel(flat[Cout: 2 * Cout].double(), y2.sum(0)) < 1e-5


def test_stem_leaves_batchnorm_statistics(dev):
    """avsr_stem357_fwd_f32s_stats: per-block partial sums of the video stem's output -> the same (mean, invstd) as the stand-alone pass."""
    B, T, H, W = 2, 5, 24, 24
    torch.manual_seed(3)
    x = torch.randn(B, T, H, W).to(dev)
    w = (0.1 * torch.randn(64, 1, 5, 7, 7)).to(dev)
    y0 = ops.stem357_fwd_f32s(x, w, B, T, H, W)
    y, part = ops.stem357_fwd_f32s(x, w, B, T, H, W, want_stats=True)
    assert torch.equal(y, y0) and torch.isfinite(part).all()
    rows = y.numel() // 64
    y2 = y.view(rows, 64).double()
    assert rel(part[:, 0].sum(0).double(), y2.sum(0)) < 1e-5 and rel(part[:, 1].sum(0).double(), (y2 * y2).sum(0)) < 1e-5
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    mean, invstd = ops.bn_finalize_parts(part, rows, 64, 1e-5, 0.1, rm, rv)
    mean0, invstd0 = ops.bn_stats_finalize(y.view(rows, 64), rows, 64, 1e-5, 0.1, torch.zeros(64, device=dev), torch.ones(64, device=dev))
    assert rel(mean, mean0) < 1e-5 and rel(invstd, invstd0) < 1e-5


def test_split8_activations(dev, twins):
    """Round 5: activations between the split-plane trunk stages travel in the split8 layout (8 hi bf16 + 8 lo bf16 per group of
    8 values, the bytes of the f32 tensor).  The BatchNorm + activation pass writes it (and reads a split8 residual), the
    split-plane convolution stages it without its in-LDS conversion pass -- bit-identical to the conversion it replaces -- and
    the cast kernel reads it back (component boundaries)."""
    torch.manual_seed(3)
    N, H, W, C, Cout = 8, 11, 11, 64, 128  # (8 x 6 x 6 = 288 output rows: three 128-row tiles, two 256-row tiles)
    rows = N * H * W
    c = torch.randn(rows, C)
    res = torch.randn(rows, C)
    mean, invstd, g, b = 0.1 * torch.randn(C), 1.0 + 0.1 * torch.rand(C), 1.0 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    z = (c - mean) * invstd * g + b

    def sp(v):  # what the layout keeps of a value: hi + lo bf16
        hi = v.bfloat16().float()
        return hi + (v - hi).bfloat16().float()

    a = ops.bn_act_fwd(c.to(dev), None, mean.to(dev), invstd.to(dev), g.to(dev), b.to(dev), rows, C, 1, out_split8=True)
    assert isinstance(a, ops.Split8) and a.dtype == torch.float32 and ops.dt(a, split8_ok=True) == 3
    want = torch.nn.functional.silu(z)
    back = ops.scale_dropout(a, torch.float32).cpu()
    assert torch.allclose(back, sp(want), rtol=2e-5, atol=2e-6) and rel(back, want) < 1e-5  # (16 significant bits)
    assert torch.equal(back, sp(back))  # what comes back is exactly representable as hi + lo
    (y, tw), = twins
    assert y.data_ptr() == a.data_ptr() and rel(tw.float(), want) < 4e-3
    assert rel(ops.scale_dropout(a, torch.float16).float(), want) < 5e-4  # (boundary to an f16 component)
    # a split8 residual + split8 output
    twins.clear()
    o = ops.bn_act_fwd(c.to(dev), a, mean.to(dev), invstd.to(dev), g.to(dev), b.to(dev), rows, C, 1, out_split8=True)
    want2 = torch.nn.functional.silu(z + back)
    assert rel(ops.scale_dropout(o, torch.float32), want2) < 1e-5
    # the convolution on the pre-split operand == the convolution that converts the f32 tile in LDS (up to the f32 rounding of
    # hi + lo when the pair is summed back for the comparison: 2^-24)
    w = 0.1 * torch.randn(Cout, C, 3, 3)
    wp = ops.conv_weight_permute_split(w.to(dev))
    x32 = back.view(N, H, W, C).to(dev)  # (values that are exactly representable in the layout)
    y_conv = ops.conv2d_fwd(x32, wp, N, H, W, C, Cout, 3, 3, 2, 1, 1, True)
    y_pre = ops.conv2d_fwd(a.view(N, H, W, C), wp, N, H, W, C, Cout, 3, 3, 2, 1, 1, True)
    assert rel(y_pre, y_conv) < 1e-6
    st = torch.empty(ops.bn_stat_tiles(N * 6 * 6), 2, Cout, device=dev)
    y_pre2 = ops.conv2d_fwd(a.view(N, H, W, C), wp, N, H, W, C, Cout, 3, 3, 2, 1, 1, True, stats=st)
    assert torch.equal(y_pre.cpu(), y_pre2.cpu())
    ref = torch.nn.functional.conv2d(back.view(N, H, W, C).double().permute(0, 3, 1, 2), w.double(), stride=2, padding=1).permute(0, 2, 3, 1)
    assert rel(y_pre, ref) < 3e-5
    # round 6: the 256-row tiles on 8 waves (knobs 21 / 22) give the same convolution, and their statistics rows -- half as many,
    # the tail of the caller's 128-row-sized buffer zeroed by the kernel -- fold to the same sums
    want_sum = y_pre.double().sum(dim=(0, 1, 2)).cpu()
    want_sq = (y_pre.double() ** 2).sum(dim=(0, 1, 2)).cpu()
    for knob, tiles in ((22, (29, 30)), (21, (27, 28))):
        Co = Cout if knob == 22 else 64
        wq = ops.conv_weight_permute_split(w[:Co].to(dev).contiguous())
        for t in tiles:
            ops.tune(knob, t)
            try:
                st2 = torch.full((ops.bn_stat_tiles(N * 6 * 6), 2, Co), float("nan"), device=dev)
                y_t = ops.conv2d_fwd(a.view(N, H, W, C), wq, N, H, W, C, Co, 3, 3, 2, 1, 1, True, stats=st2)
            finally:
                ops.tune(knob, 0)
            assert rel(y_t, y_pre[..., :Co]) < 1e-6, t
            got = st2.double().sum(0).cpu()
            assert torch.allclose(got[0], want_sum[:Co], rtol=1e-5, atol=1e-4) and torch.allclose(got[1], want_sq[:Co], rtol=1e-5, atol=1e-4), t
    # a consumer that cannot read the layout fails loudly
    with pytest.raises(TypeError):
        ops.bn_stats(a, rows, C)


# ---------------------------------------------------------------------------------------------------------------- module level
POLICIES = {
    "default": None,
    "default-casts": None,  # no producer-side twins (what tensors below functional._TWIN_MIN get): save-time casts, same layouts
    "encoder-only": {"encoder": "f16"},
    "one-plane": {"encoder": "f16", "decoder": "f16", "trunk3": "f16", "trunk4": "f16", "atrunk3": "f16", "atrunk4": "f16"},  # round 4's default
    "all-split": {},
}


@pytest.mark.parametrize("modality,policy", [("video", "default"), ("video", "default-casts"), ("video", "all-split"), ("video", "one-plane"),
                                             ("audio", "default")])
def test_e2e_small_mixed_mode(dev, modality, policy, monkeypatch):
    """Small E2E instance in the mixed mode against the fp32 oracle: losses inside the north-star bound (1e-3; f16 operands
    deliver ~1e-4 here), gradients aligned, every saved activation of the f16 components picked up as a producer-side twin, and
    -- policy {} -- bit-identical to the hpf mode (everything on split planes)."""
    from oracle import avsr_oracle as O
    from synth import synth_state_dict
    from test_modules import no_dropout, synth_batch

    from auto_avsr_amd.e2e import E2E

    if POLICIES[policy] is not None:
        monkeypatch.setattr(AF, "MIXED_POLICY", POLICIES[policy])
    from auto_avsr_amd import functional_frontend as FF

    if policy == "all-split":  # bit-identity with hpf below: BatchNorm statistics from the stand-alone pass, as hpf takes them
        monkeypatch.setattr(FF, "_FUSE_BN_STATS", False)
        monkeypatch.setattr(FF, "_PRESPLIT", False)  # (and plain f32 activations between the trunk stages, as in hpf)
    trace = []
    monkeypatch.setattr(ops, "TRACE", trace)
    torch.manual_seed(0)
    odim = 72
    m = no_dropout(E2E(odim, modality, adim=128, aheads=2, eunits=256, elayers=2, dunits=256, dlayers=2, cnn_module_kernel=7))
    sd = synth_state_dict(m.state_dict(), 13)
    m.load_state_dict(sd, strict=True)
    m.to(dev).train()
    x, lengths, y = synth_batch(modality, 2, 9, 4, odim, seed=8)
    osd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
    (loss_r, ctc_r, att_r, acc_r), _ = O.e2e_forward(osd, x, lengths, y, modality=modality, heads=2)
    loss_r.backward()
    AF.invalidate_weight_cache()
    monkeypatch.setattr(AF, "_TWIN_MIN", (1 << 60) if policy == "default-casts" else 0)
    AF._twin_stats.update(made=0, used=0, cast=0)
    with AF.numerics("mixed"):
        assert AF.mode() == "mixed"
        loss, loss_ctc, loss_att, acc = m(x.to(dev), lengths.to(dev), y.to(dev))
        loss.backward()
    assert AF.mode() == "bf16" and ops.TWIN is None and not AF._state["f16"]
    assert abs(float(loss_ctc) - float(ctc_r)) < 1e-3 * abs(float(ctc_r))
    assert abs(float(loss_att) - float(att_r)) < 1e-3 * abs(float(att_r))
    assert acc == acc_r
    names = [t[0] for t in trace]
    if policy != "all-split":  # split-plane trunk stages 1 - 2 (2-D and 1-D ResNet alike): statistics come out of the convolution epilogue
        n_stem = int(modality == "video")  # (the video stem kernel leaves its statistics too; the audio stem is a tiled conv1d)
        assert names.count("avsr_conv2d_f32s_stats") == 9 and names.count("avsr_stem357_fwd_f32s_stats") == n_stem \
            and names.count("avsr_bn_finalize_parts") == 9 + n_stem, \
            (names.count("avsr_conv2d_f32s_stats"), names.count("avsr_bn_finalize_parts"))
    else:
        assert "avsr_conv2d_f32s_stats" not in names
    st = dict(AF._twin_stats)
    if policy == "default-casts":
        assert st["made"] == 0 and st["cast"] > 20, st
    else:
        assert st["used"] > 20 and st["cast"] <= 8, st
    cos = []
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32, k
        a, b = p.grad.double().flatten().cpu(), osd[k].grad.double().flatten()
        if b.norm() > 1e-4 * max(1.0, float(osd[k].double().norm())):
            cos.append((float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), k))
    assert min(cos)[0] > 0.9, sorted(cos)[:5]
    if policy == "all-split":
        m.load_state_dict(sd, strict=True)
        AF.invalidate_weight_cache()
        with AF.numerics("hpf"):
            ref = [float(v) for v in m(x.to(dev), lengths.to(dev), y.to(dev))[:3]]
        assert [float(loss), float(loss_ctc), float(loss_att)] == ref
    AF.invalidate_weight_cache()


def test_side_weight_caches_follow_raw_pointer_updates(dev):
    """Round-3 advisor finding: an optimizer that updates weights through raw pointers (no tensor version bump) followed by a
    bf16-mode refresh must not leave the split8 / f16 side copies looking fresh."""
    torch.manual_seed(4)
    AF.invalidate_weight_cache()
    lin = torch.nn.Linear(64, 128).to(dev)
    x = torch.randn(16, 64, device=dev)

    def run(mode):
        with AF.numerics(mode), torch.no_grad():
            if mode == "mixed":
                with AF.component("encoder"):
                    return AF.linear(x, lin.weight, lin.bias, out_dtype=torch.float32).clone()
            return AF.linear(x, lin.weight, lin.bias, out_dtype=torch.float32).clone()

    y_split, y_f16 = run("precise"), run("mixed")
    lin.weight.data.view(-1)[:].mul_(2.0)  # in place through .data: _version unchanged, like the fused optimizer's raw-pointer update
    AF.note_optimizer_step(False)
    run("bf16")  # a bf16-mode use in between refreshes the bf16 copies and clears the dirty flag
    b = lin.bias.detach()
    for y_old, mode, tol in ((y_split, "precise", 1e-4), (y_f16, "mixed", 2e-3)):
        y_new = run(mode)
        assert rel(y_new - b, 2.0 * (y_old - b)) < tol, mode
    AF.invalidate_weight_cache()
