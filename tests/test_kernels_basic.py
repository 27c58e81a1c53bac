"""Kernel-level parity: LayerNorm and the MFMA GEMM family against plain fp32/fp64 torch math.
Each test runs on the host emulator build (CPU suite) and on the gfx950 build (-m gpu)."""
import pytest
import torch

from auto_avsr_amd import ops

DT = {0: torch.float32, 1: torch.bfloat16}


def test_layernorm_fwd_bwd(dev):
    torch.manual_seed(0)
    rows, cols = 37, 768
    x, g, b = torch.randn(rows, cols), torch.randn(cols), torch.randn(cols)
    dy, dres = torch.randn(rows, cols), torch.randn(rows, cols)
    xr, gr, br = (t.clone().requires_grad_() for t in (x, g, b))
    ref = torch.nn.functional.layer_norm(xr, (cols,), gr, br, 1e-12)
    ref.backward(dy)
    xd, gd, bd, dyd, dresd = (t.to(dev) for t in (x, g, b, dy, dres))
    y, mean, rstd = ops.layernorm_fwd(xd, gd, bd, torch.float32)
    assert (y.cpu() - ref).abs().max() < 1e-5
    dg, db = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    dx = ops.layernorm_bwd(dyd, xd, gd, mean, rstd, dg, db, dres=dresd)
    assert (dx.cpu() - dres - xr.grad).abs().max() < 1e-4
    assert (dg.cpu() - gr.grad).abs().max() < 1e-4
    assert (db.cpu() - br.grad).abs().max() < 1e-4
    yb, _, _ = ops.layernorm_fwd(xd, gd, bd, torch.bfloat16)
    assert (yb.float().cpu() - ref).abs().max() < 0.05
    # bf16 dy path
    dg.zero_(); db.zero_()
    dx2 = ops.layernorm_bwd(dyd.bfloat16(), xd, gd, mean, rstd, dg, db)
    assert (dx2.cpu() - xr.grad).abs().max() < 0.05


@pytest.mark.parametrize("rows", [37, 1100])
def test_layernorm_bwd_fused_prologue(dev, rows):
    """avsr_layernorm_bwd gout / gsum: the consumer Linear's backward prologue (bf16(alpha * dropout(dx)) and its column
    sums) produced in the LayerNorm backward pass must equal what the separate avsr_cast_transpose_colsum pass makes of the
    same dx -- same dropout stream, same rounding -- and leave dx / dgamma / dbeta untouched.  rows = 1100: two rows per
    wave."""
    torch.manual_seed(1)
    cols = 768
    x, g, b = torch.randn(rows, cols), torch.randn(cols), torch.randn(cols)
    dy, dres = torch.randn(rows, cols), torch.randn(rows, cols)
    xd, gd, bd, dyd, dresd = (t.to(dev) for t in (x, g, b, dy.bfloat16(), dres))
    _, mean, rstd = ops.layernorm_fwd(xd, gd, bd, torch.float32)
    dg0, db0 = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
    dx0 = ops.layernorm_bwd(dyd, xd, gd, mean, rstd, dg0, db0, dres=dresd)
    for (alpha, p, seed) in ((0.5, 0.1, 1234567), (1.0, 0.0, 0)):
        cs0 = torch.zeros(cols, device=dev)
        want, _ = ops.cast_transpose_colsum(dx0, rows, cols, want_dst=True, want_T=False, colsum=cs0, alpha=alpha, drop_p=p,
                                            seed=seed)
        dg1, db1, cs1 = torch.zeros(cols, device=dev), torch.zeros(cols, device=dev), torch.zeros(cols, device=dev)
        gout = torch.empty(rows, cols, dtype=torch.bfloat16, device=dev)
        dx1 = ops.layernorm_bwd(dyd, xd, gd, mean, rstd, dg1, db1, dres=dresd, gout=gout, gsum=cs1, alpha=alpha, drop_p=p,
                                seed=seed)
        assert torch.equal(dx1, dx0)
        assert torch.equal(gout, want)
        assert (cs1 - cs0).abs().max() <= 1e-4 * max(1.0, float(cs0.abs().max()))
        assert (dg1 - dg0).abs().max() <= 1e-4 * max(1.0, float(dg0.abs().max()))
        assert (db1 - db0).abs().max() <= 1e-4 * max(1.0, float(db0.abs().max()))
        if p > 0:
            frac = float((gout == 0).float().mean())
            assert 0.05 < frac < 0.15


def _pad(t):
    r, c = t.shape
    c8 = (c + 7) // 8 * 8 + 8
    buf = torch.zeros(r, c8, dtype=t.dtype)
    buf[:, :c] = t
    return buf, c8


def _run_gemm(dev, layout, M, N, K, adt, bdt, precise, tile, cdt=0, split=1, epi=False):
    A, B = torch.randn(M, K), torch.randn(N, K)
    Aq, Bq = A.to(DT[adt]), B.to(DT[bdt])
    As, lda = _pad(Aq if layout != 2 else Aq.t().contiguous())
    Bs, ldb = _pad(Bq if layout == 0 else Bq.t().contiguous())
    ldc = N + 3
    C = torch.zeros(M, ldc, dtype=DT[cdt], device=dev)
    ref = Aq.double() @ Bq.double().t()
    kw = {}
    if epi:
        bias, resid = torch.randn(N), torch.randn(M, N)
        ref = torch.relu(ref + bias.double()) * 0.5 + resid.double()
        kw = dict(bias=bias.to(dev), act=1, alpha=0.5, resid=resid.to(dev), ldr=N)
    ops.gemm(layout, As.to(dev), lda, Bs.to(dev), ldb, M, N, K, C, ldc, precise=bool(precise),
             accumulate=split > 1, split_k=split, force_tile=tile, **kw)
    out = C.cpu()[:, :N].double()
    assert C.cpu()[:, N:].abs().max() == 0, "wrote outside the logical columns"
    return ((out - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("dts", [(1, 1, 0), (0, 1, 0), (1, 0, 0), (0, 0, 0), (0, 0, 1)])
@pytest.mark.parametrize("tile", [64, 128])
def test_gemm_layouts_dtypes(dev, layout, dts, tile):
    torch.manual_seed(layout * 10 + tile)
    adt, bdt, precise = dts
    M, N, K = (150, 70, 200) if tile == 64 else (200, 130, 136)
    err = _run_gemm(dev, layout, M, N, K, adt, bdt, precise, tile)
    # bf16-stored operands multiply exactly in f32; f32 operands are rounded to bf16 in-kernel
    tol = 2e-5 if precise else (1e-2 if (adt == 0 or bdt == 0) else 2e-6)
    assert err < tol, err


def test_gemm_epilogue_split_ragged(dev):
    torch.manual_seed(3)
    assert _run_gemm(dev, 0, 100, 96, 64, 1, 1, 0, 64, cdt=0, epi=True) < 2e-6
    assert _run_gemm(dev, 0, 100, 96, 64, 1, 1, 0, 64, cdt=1, epi=True) < 1e-2
    assert _run_gemm(dev, 2, 96, 80, 700, 1, 1, 0, 64, split=4) < 2e-6
    assert _run_gemm(dev, 1, 64, 64, 203, 1, 1, 0, 64) < 2e-6
    assert _run_gemm(dev, 0, 1, 5, 8, 1, 1, 0, 0) < 2e-6


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("shape", [(150, 70, 192), (200, 130, 64), (257, 300, 448)])
def test_gemm_fast_nt(dev, tile, shape):
    """LDS-DMA / swizzled-LDS / counted-vmcnt NT kernel: exact products of bf16 operands, all tiles, ragged M/N,
    pipeline depths shorter and longer than the ring."""
    M, N, K = shape
    torch.manual_seed(M + tile)
    A, B = torch.randn(M, K).bfloat16(), torch.randn(N, K).bfloat16()
    ref = A.double() @ B.double().t()
    C = torch.zeros(M, N + 3, device=dev)
    ops.gemm_bf16_nt(A.to(dev), K, B.to(dev), K, M, N, K, C, N + 3, tile=tile)
    assert ((C.cpu()[:, :N].double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert C.cpu()[:, N:].abs().max() == 0
    # split-K + epilogue (bias, relu, alpha, bf16 residual) + bf16 output
    bias, resid = torch.randn(N), torch.randn(M, N).bfloat16()
    C2 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm_bf16_nt(A.to(dev), K, B.to(dev), K, M, N, K, C2, N, bias=bias.to(dev), act=1, alpha=0.5,
                     resid=resid.to(dev), ldr=N, tile=tile)
    ref2 = torch.relu(ref + bias.double()) * 0.5 + resid.double()
    assert ((C2.cpu().double() - ref2).abs().max() / ref2.abs().max()) < 1e-2
    # column sums of the stored values from the epilogue (bias gradient of the layer whose output gradient C is)
    cs = torch.full((N,), 3.0, device=dev)
    ops.gemm_bf16_nt(A.to(dev), K, B.to(dev), K, M, N, K, C2, N, bias=bias.to(dev), act=1, alpha=0.5,
                     resid=resid.to(dev), ldr=N, tile=tile, colsum=cs)
    assert ((cs.cpu().double() - 3.0 - ref2.sum(0)).abs().max() / ref2.sum(0).abs().max()) < 1e-4
    C3 = torch.zeros(M, N, device=dev)
    ops.gemm_bf16_nt(A.to(dev), K, B.to(dev), K, M, N, K, C3, N, accumulate=True, split_k=3, tile=tile)
    assert ((C3.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("shape", [(150, 70, 192), (200, 130, 64), (257, 300, 448)])
def test_gemm_split_f32_nt(dev, tile, shape):
    """Precise-mode NT kernel on the LDS-DMA ring (gemm_split.hip): f32 operands, split hi / lo bf16 planes formed on the
    fragments, three MFMAs per product -- f32-class accuracy vs fp64 (same bound as the generic precise kernel), pitched
    operands, all tiles, ragged M / N, ring depths shorter and longer than the k loop, epilogue, split-K, column sums."""
    M, N, K = shape
    torch.manual_seed(M + tile)
    A, B = torch.randn(M, K), torch.randn(N, K)
    ref = A.double() @ B.double().t()
    Ap = torch.zeros(M, K + 8)
    Ap[:, :K] = A
    C = torch.zeros(M, N + 3, device=dev)
    ops.gemm_f32s_nt(Ap.to(dev), K + 8, B.to(dev), K, M, N, K, C, N + 3, tile=tile)
    assert ((C.cpu()[:, :N].double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert C.cpu()[:, N:].abs().max() == 0
    # the generic precise kernel does the same arithmetic: results agree to summation order
    Cg = torch.zeros(M, N, device=dev)
    ops.gemm(0, A.to(dev), K, B.to(dev), K, M, N, K, Cg, N, precise=True)
    assert ((C.cpu()[:, :N] - Cg.cpu()).abs().max() / ref.abs().max()) < 2e-6
    bias, resid = torch.randn(N), torch.randn(M, N)
    C2 = torch.zeros(M, N, device=dev)
    cs = torch.full((N,), 3.0, device=dev)
    ops.gemm_f32s_nt(A.to(dev), K, B.to(dev), K, M, N, K, C2, N, bias=bias.to(dev), act=1, alpha=0.5,
                     resid=resid.to(dev), ldr=N, tile=tile, colsum=cs)
    ref2 = torch.relu(ref + bias.double()) * 0.5 + resid.double()
    assert ((C2.cpu().double() - ref2).abs().max() / ref2.abs().max()) < 2e-5
    assert ((cs.cpu().double() - 3.0 - ref2.sum(0)).abs().max() / ref2.sum(0).abs().max()) < 1e-4
    C3 = torch.zeros(M, N, device=dev)
    ops.gemm_f32s_nt(A.to(dev), K, B.to(dev), K, M, N, K, C3, N, accumulate=True, split_k=3, tile=tile)
    assert ((C3.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-5
    # B pre-split into hi / lo planes (split8 layout, what the weights of the precise forward pass use): the same products
    Bs = ops.split_pack(B.to(dev))
    assert isinstance(Bs, ops.Split8) and Bs.shape == B.shape
    C4 = torch.zeros(M, N + 3, device=dev)
    ops.gemm_f32s_nt(Ap.to(dev), K + 8, Bs, K, M, N, K, C4, N + 3, tile=tile)
    assert torch.equal(C4.cpu(), C.cpu()), "pre-split B must give bit-identical results"


def test_transpose_cast(dev):
    torch.manual_seed(9)
    x = torch.randn(150, 70)
    t = ops.transpose_cast(x.to(dev), 150, 70)
    assert t.shape == (70, 192)
    assert (t.cpu()[:, :150].float() - x.t().bfloat16().float()).abs().max() == 0 and t.cpu()[:, 150:].abs().max() == 0
    xb = torch.randn(33, 136).bfloat16()
    t2 = ops.transpose_cast(xb.to(dev), 33, 136)
    assert (t2.cpu()[:, :33] == xb.t()).all()


def test_cast_transpose_colsum(dev):
    torch.manual_seed(10)
    R, C = 150, 136
    x = torch.randn(R, C)
    cs = torch.zeros(C, device=dev)
    dst, dstT = ops.cast_transpose_colsum(x.to(dev), R, C, want_dst=True, colsum=cs, alpha=0.5)
    ref = (0.5 * x).bfloat16()
    assert (dst.cpu() == ref).all()
    assert (dstT.cpu()[:, :R] == ref.t()).all() and dstT.cpu()[:, R:].abs().max() == 0
    assert (cs.cpu() - ref.float().sum(0)).abs().max() < 1e-3
    # dropout: same mask as scale_dropout with the same (seed, index)
    d2, _ = ops.cast_transpose_colsum(x.to(dev), R, C, want_dst=True, want_T=False, drop_p=0.3, seed=99)
    ref2 = ops.scale_dropout(x.to(dev), torch.bfloat16, drop_p=0.3, seed=99)
    assert (d2.cpu() == ref2.cpu()).all()


@pytest.mark.parametrize("shape", [(72, 136, 150), (64, 64, 64), (200, 72, 1000)])
def test_gemm_bf16_tn_transpose_read(dev, shape):
    """Weight-gradient kernel: k-major LDS tiles + ds_read_b64_tr_b16 fragments, ragged K / M / N, split-K."""
    M, N, K = shape
    torch.manual_seed(M + K)
    A, B = torch.randn(K, M).bfloat16(), torch.randn(K, N).bfloat16()
    ref = A.double().t() @ B.double()
    C = torch.zeros(M, N + 5, device=dev)
    ops.gemm_bf16_tn(A.to(dev), M, B.to(dev), N, M, N, K, C, N + 5)
    assert ((C.cpu()[:, :N].double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert C.cpu()[:, N:].abs().max() == 0
    C2 = torch.zeros(M, N, device=dev)
    cs = torch.zeros(M, device=dev)
    ops.gemm_bf16_tn(A.to(dev), M, B.to(dev), N, M, N, K, C2, N, accumulate=True, split_k=3, colsum_a=cs)
    assert ((C2.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6
    # the bias gradient on the side: column sums of A (= dY) from the staged tiles, across the k-splits
    want = A.double().sum(0)
    assert ((cs.cpu().double() - want).abs().max() / want.abs().max().clamp_min(1.0)) < 1e-5


@pytest.mark.parametrize("tile", [1, 7])
@pytest.mark.parametrize("split", [1, 2])
def test_gemm_pair_nt_tn(dev, tile, split):
    """Data-gradient (NT) + weight-gradient (TN) GEMM of a Linear backward as ONE launch (csrc/gemm_pair.hip): the grid
    holds the tiles of both problems; results equal the two separate launches bit for bit (same tile kernels)."""
    torch.manual_seed(tile * 10 + split)
    rows, n_out, n_in = 200, 136, 192  # dX[rows, n_in] = dY[rows, n_out(64-padded K)] W ; dW[n_out, n_in] = dY^T X
    Kp = 192
    dy = torch.zeros(rows, Kp)
    dy[:, :n_out] = torch.randn(rows, n_out)
    dy = dy.bfloat16().to(dev)
    wT = torch.zeros(n_in, Kp)
    wT[:, :n_out] = torch.randn(n_in, n_out)
    wT = wT.bfloat16().to(dev)
    x = torch.randn(rows, n_in).bfloat16().to(dev)
    bias = torch.randn(n_in, device=dev)

    def nt(out, cs):
        ops.gemm_bf16_nt(dy, Kp, wT, Kp, rows, n_in, Kp, out, n_in, bias=bias, act=1, tile=tile, colsum=cs)

    csa = []

    def tn(out):
        csa.append(torch.zeros(n_out, device=dev))
        ops.gemm_bf16_tn(dy, Kp, x, n_in, n_out, n_in, rows, out, n_in, accumulate=split > 1, split_k=split, colsum_a=csa[-1])

    dx0, cs0, dw0 = torch.zeros(rows, n_in, device=dev, dtype=torch.bfloat16), torch.zeros(n_in, device=dev), torch.zeros(n_out, n_in, device=dev)
    nt(dx0, cs0)
    tn(dw0)
    ref_dw = dy.cpu().double()[:, :n_out].t() @ x.cpu().double()
    assert ((dw0.cpu().double() - ref_dw).abs().max() / ref_dw.abs().max()) < 2e-6
    dx1, cs1, dw1 = torch.zeros_like(dx0), torch.zeros_like(cs0), torch.zeros_like(dw0)
    with ops.paired():
        tn(dw1)
        nt(dx1, cs1)
    assert torch.equal(dx1, dx0) and torch.equal(dw1, dw0)
    assert (cs1 - cs0).abs().max() <= 1e-4 * cs0.abs().max()  # atomics: summation order differs
    # a pair holding only one of the two, and calls the pair cannot hold (second NT, split-K NT) launch as usual
    dx2, dw2, dx3 = torch.zeros_like(dx0), torch.zeros_like(dw0), torch.zeros(rows, n_in, device=dev)
    with ops.paired():
        nt(dx2, None)
        ops.gemm_bf16_nt(dy, Kp, wT, Kp, rows, n_in, Kp, dx3, n_in, accumulate=True, split_k=3, tile=tile)
    with ops.paired():
        tn(dw2)
    with ops.paired():
        pass
    assert torch.equal(dx2, dx0) and torch.equal(dw2, dw0)
    want_cs = dy.cpu().double()[:, :n_out].sum(0)  # colsum(dY): stand-alone launch, paired launch, pair of one
    for c in csa:
        assert ((c.cpu().double() - want_cs).abs().max() / want_cs.abs().max().clamp_min(1.0)) < 1e-5
    ref_dx = dy.cpu().double() @ wT.cpu().double().t()
    assert ((dx3.cpu().double() - ref_dx).abs().max() / ref_dx.abs().max()) < 2e-6


@pytest.mark.parametrize("shape", [(200, 136, 192), (333, 256, 320), (130, 128, 128)])
def test_gemm_pair_tn_128_tiles(dev, shape):
    """Round 5: the weight-gradient half of a paired launch on 128 x 128 tiles (gemm_tn_kernel.h TMS = TNS = 2; measured slower
    than the 64 x 64 tiles on the MI355X and therefore behind tuning knob 19 = 2): ragged M / N / K edges, the bias-gradient column sums of both 64-column sub-tiles,
    results against float64 and against the 64 x 64 tiles."""
    rows, n_out, n_in = shape
    torch.manual_seed(rows)
    Kp = (n_out + 63) // 64 * 64
    dy = torch.zeros(rows, Kp)
    dy[:, :n_out] = torch.randn(rows, n_out)
    dy = dy.bfloat16().to(dev)
    wT = torch.randn(n_in, Kp).bfloat16().to(dev)
    x = torch.randn(rows, n_in).bfloat16().to(dev)
    ref_dw = dy.cpu().double()[:, :n_out].t() @ x.cpu().double()
    ref_dx = dy.cpu().double() @ wT.cpu().double().t()
    want_cs = dy.cpu().double()[:, :n_out].sum(0)
    res = {}
    for knob in (0, 2):  # 64 x 64 tiles (the default), 128 x 128 tiles
        ops.tune(19, knob)
        try:
            dx = torch.zeros(rows, n_in, device=dev, dtype=torch.bfloat16)
            dw = torch.full((n_out, n_in), 7.0, device=dev)  # (overwritten, not accumulated into)
            cs = torch.zeros(n_out, device=dev)
            with ops.paired():
                ops.gemm_bf16_tn(dy, Kp, x, n_in, n_out, n_in, rows, dw, n_in, accumulate=False, split_k=1, colsum_a=cs)
                ops.gemm_bf16_nt(dy, Kp, wT, Kp, rows, n_in, Kp, dx, n_in, tile=1)
        finally:
            ops.tune(19, 0)
        assert ((dw.cpu().double() - ref_dw).abs().max() / ref_dw.abs().max()) < 2e-6, knob
        assert ((dx.cpu().double() - ref_dx).abs().max() / ref_dx.abs().max()) < 1e-2, knob
        assert ((cs.cpu().double() - want_cs).abs().max() / want_cs.abs().max().clamp_min(1.0)) < 1e-5, knob
        res[knob] = dw.cpu()
    assert (res[0] - res[2]).abs().max() <= 1e-5 * res[0].abs().max()  # (same products, f32 sums in another order)


def test_gemm_pair_tn_deep_fragment_pipeline(dev):
    """Tuning knob 20 = 1: the weight-gradient tile requests the transpose reads of all four k-steps of a staged tile up front
    (gemm_tn_kernel.h DEPTH = 4).  Same products in the same order: results equal the default's bit for bit."""
    torch.manual_seed(5)
    rows, n_out, n_in = 333, 192, 256
    dy = torch.randn(rows, n_out).bfloat16().to(dev)
    wT = torch.randn(n_in, n_out).bfloat16().to(dev)
    x = torch.randn(rows, n_in).bfloat16().to(dev)
    out = {}
    for knob in (0, 1):
        ops.tune(20, knob)
        try:
            dx = torch.zeros(rows, n_in, device=dev, dtype=torch.bfloat16)
            dw = torch.zeros(n_out, n_in, device=dev)
            with ops.paired():
                ops.gemm_bf16_tn(dy, n_out, x, n_in, n_out, n_in, rows, dw, n_in, accumulate=True, split_k=2)
                ops.gemm_bf16_nt(dy, n_out, wT, n_out, rows, n_in, n_out, dx, n_in, tile=1)
        finally:
            ops.tune(20, 0)
        out[knob] = (dx.cpu(), dw.cpu())
    ref = dy.cpu().double().t() @ x.cpu().double()
    assert ((out[1][1].double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert torch.equal(out[0][0], out[1][0]) and (out[0][1] - out[1][1]).abs().max() <= 1e-5 * out[0][1].abs().max()
