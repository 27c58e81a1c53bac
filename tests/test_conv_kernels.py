"""Implicit-GEMM convolutions and pooling (front-ends) vs torch conv2d / conv3d / pooling autograd."""
import pytest
import torch
import torch.nn.functional as F

from auto_avsr_amd import ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, K, stride, pad
    (3, 11, 9, 16, 24, 3, 1, 1),
    (2, 12, 10, 64, 72, 3, 2, 1),
    (2, 11, 11, 32, 40, 1, 2, 0),
    (2, 1, 37, 16, 32, 3, 2, 1),   # 1-D (audio trunk) as an H = 1 image: KH = 1
    (2, 9, 10, 64, 128, 3, 1, 1),  # channel counts that take the LDS-DMA bf16 kernel in bf16 mode
    (2, 10, 9, 128, 64, 3, 2, 1),
    (3, 7, 7, 64, 128, 1, 2, 0),
    (2, 11, 7, 64, 64, 3, 2, 1),   # odd extents: the residue classes of the strided data gradient differ in size
])
@pytest.mark.parametrize("precise", [True, False])
def test_conv2d_fwd_dgrad_wgrad(dev, cfg, precise):
    N, H, W, Cin, Cout, K, s, p = cfg
    torch.manual_seed(N * 7 + K)
    KH, ph = (1, 0) if H == 1 else (K, p)
    dtype = torch.float32 if precise else torch.bfloat16
    x = torch.randn(N, Cin, H, W).to(dtype).float().requires_grad_()
    w = (torch.randn(Cout, Cin, KH, K) / (Cin * KH * K) ** 0.5).requires_grad_()
    wq = w.detach().to(dtype).float().requires_grad_()
    y_ref = F.conv2d(x, wq, stride=s, padding=(ph, p))
    dy = torch.randn_like(y_ref).to(dtype).float()
    y_ref.backward(dy)
    tol = 2e-5 if precise else 2e-2
    xd = nhwc(x.detach()).to(dtype).to(dev)
    wp = ops.conv_weight_permute(w.detach().to(dev), dtype)
    y = ops.conv2d_fwd(xd, wp, N, H, W, Cin, Cout, KH, K, s, ph, p, precise)
    assert (y.float().cpu() - nhwc(y_ref.detach())).abs().max() < tol * max(1.0, y_ref.abs().max().item())
    dyd = nhwc(dy).to(dtype).to(dev)
    wpd = ops.conv_weight_permute(w.detach().to(dev), dtype, to_dgrad=True)
    dx = ops.conv2d_dgrad(dyd, wpd, None, N, H, W, Cin, Cout, KH, K, s, ph, p, precise)
    assert (dx.float().cpu() - nhwc(x.grad)).abs().max() < tol * max(1.0, x.grad.abs().max().item())
    res = torch.randn(N, H, W, Cin).to(dtype)
    dx2 = ops.conv2d_dgrad(dyd, wpd, res.to(dev), N, H, W, Cin, Cout, KH, K, s, ph, p, precise)
    assert (dx2.float().cpu() - nhwc(x.grad) - res.float()).abs().max() < 2 * tol * max(1.0, x.grad.abs().max().item())
    dwp = ops.conv2d_wgrad(dyd, xd, N, H, W, Cin, Cout, KH, K, s, ph, p, precise)
    dw = ops.conv_weight_unpermute(dwp, wq.shape)
    assert (dw.cpu() - wq.grad).abs().max() < tol * max(1.0, wq.grad.abs().max().item())


@pytest.mark.parametrize("tile", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("cfg", [(5, 9, 10, 64, 128, 3, 1, 1), (4, 11, 12, 128, 64, 3, 2, 1), (9, 7, 7, 64, 192, 1, 2, 0)])
def test_conv2d_bf16_tiles(dev, cfg, tile):
    """Every tile / wave-count variant of the tuned bf16 convolution kernel gives the same forward and data gradient."""
    N, H, W, Cin, Cout, K, s, p = cfg
    torch.manual_seed(tile)
    x = torch.randn(N, Cin, H, W).bfloat16().float()
    w = (torch.randn(Cout, Cin, K, K) / (Cin * K * K) ** 0.5)
    wq = w.bfloat16().float()
    x.requires_grad_()
    y_ref = F.conv2d(x, wq, stride=s, padding=p)
    dy = torch.randn_like(y_ref).bfloat16().float()
    y_ref.backward(dy)
    ops.tune(0, tile)
    try:
        xd = nhwc(x.detach()).bfloat16().to(dev)
        y = ops.conv2d_fwd(xd, ops.conv_weight_permute(w.to(dev), torch.bfloat16), N, H, W, Cin, Cout, K, K, s, p, p, False)
        res = torch.randn(N, H, W, Cin).bfloat16()
        dx = ops.conv2d_dgrad(nhwc(dy).bfloat16().to(dev), ops.conv_weight_permute(w.to(dev), torch.bfloat16, to_dgrad=True),
                              res.to(dev), N, H, W, Cin, Cout, K, K, s, p, p, False)
    finally:
        ops.tune(0, 0)
    assert (y.float().cpu() - nhwc(y_ref.detach())).abs().max() < 2e-2 * max(1.0, y_ref.abs().max().item())
    assert (dx.float().cpu() - nhwc(x.grad) - res.float()).abs().max() < 4e-2 * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("cfg", [(5, 9, 10, 64, 128, 3, 1, 1), (4, 11, 12, 128, 64, 3, 2, 1), (9, 7, 7, 64, 192, 1, 2, 0),
                                 (2, 1, 37, 64, 64, 3, 2, 1)])
def test_conv2d_f32_split_forward(dev, cfg):
    """Precise-mode forward convolution on the LDS-DMA ring (gemm_split.hip CV = 1) vs torch conv2d in f32 and vs the generic
    precise kernel it replaces (AVSR_SPLIT_FAST switch)."""
    N, H, W, Cin, Cout, K, s, p = cfg
    torch.manual_seed(N)
    KH, ph = (1, 0) if H == 1 else (K, p)
    x = torch.randn(N, Cin, H, W)
    w = torch.randn(Cout, Cin, KH, K) / (Cin * KH * K) ** 0.5
    y_ref = F.conv2d(x.double(), w.double(), stride=s, padding=(ph, p)).float()
    xd = nhwc(x).to(dev)
    wp = ops.conv_weight_permute(w.to(dev), torch.float32)
    assert ops.SPLIT_FAST
    y = ops.conv2d_fwd(xd, wp, N, H, W, Cin, Cout, KH, K, s, ph, p, True)
    ops.SPLIT_FAST = False
    try:
        yg = ops.conv2d_fwd(xd, wp, N, H, W, Cin, Cout, KH, K, s, ph, p, True)
    finally:
        ops.SPLIT_FAST = True
    scale = max(1.0, y_ref.abs().max().item())
    assert (y.cpu() - nhwc(y_ref)).abs().max() < 2e-5 * scale
    assert (y.cpu() - yg.cpu()).abs().max() < 2e-6 * scale
    ys = ops.conv2d_fwd(xd, ops.conv_weight_permute_split(w.to(dev)), N, H, W, Cin, Cout, KH, K, s, ph, p, True)
    assert torch.equal(ys.cpu(), y.cpu()), "pre-split weights must give bit-identical results"


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, stride -- trunk geometries in miniature: row bands (tall images), several whole images per
    # tile (small images), ragged last band / last image group, stride 2 with odd and even extents
    (3, 22, 22, 64, 64, 1), (5, 11, 11, 128, 64, 1), (7, 6, 6, 64, 128, 1), (17, 3, 3, 64, 64, 1),
    (3, 22, 22, 64, 128, 2), (4, 11, 11, 64, 64, 2), (5, 6, 6, 128, 64, 2), (2, 9, 13, 64, 64, 1), (2, 7, 10, 64, 64, 2),
])
@pytest.mark.parametrize("partial", [True, False])
@pytest.mark.parametrize("variant", ["thin", "thin-long", "fat-long", "thin8-long"])
def test_conv3x3_wgrad_direct(dev, cfg, partial, variant):
    """Dedicated 3x3 weight-gradient kernel (shifted LDS views of one padded patch, fragment requests pipelined across the
    16-pixel k-steps) vs torch autograd.  Variants (avsr_tune knob 16): thin = 4 waves of 32 co x 32 ci; fat = 4 waves of
    64 co x 32 ci in two k groups that share each staged tile (3-stage ring, LDS hand-over of the accumulators); thin8 = 8 thin
    waves in two k groups.  "-long": ONE block per (co, ci) pair, so that the staging ring and the k-step pipeline run over
    every tile of the problem."""
    ops.tune(16, {"thin": 1, "fat": 2, "thin8": 3}[variant.split("-")[0]])
    ops.tune(15, 1 if variant.endswith("long") else 0)
    try:
        _wgrad_direct(dev, cfg, partial)
    finally:
        ops.tune(16, 0)
        ops.tune(15, 0)


def _wgrad_direct(dev, cfg, partial):
    N, H, W, Cin, Cout, s = cfg
    torch.manual_seed(H * 31 + N)
    x = torch.randn(N, Cin, H, W).bfloat16().float()
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    y = F.conv2d(x, w, stride=s, padding=1)
    dy = torch.randn_like(y).bfloat16().float()
    y.backward(dy)
    OH, OW = y.shape[2], y.shape[3]
    dwp = torch.zeros(Cout, 9 * Cin, dtype=torch.float32, device=dev)
    dyd, xd = nhwc(dy).bfloat16().to(dev), nhwc(x).bfloat16().to(dev)
    nws = ops.call("avsr_conv3x3_wgrad_workspace_bytes", N, H, W, Cin, Cout, s)
    assert nws > 0 and nws % (Cout * 9 * Cin * 4) == 0
    ws = torch.full((nws // 4,), float("nan"), device=dev) if partial else None  # every partial slot must be written
    if partial:
        dwp.fill_(float("nan"))  # partial mode overwrites
    ops.call("avsr_conv3x3_wgrad_bf16", ops._ptr(dyd), ops._ptr(xd), ops._ptr(dwp), ops._ptr(ops.zero_page(dev)),
             ops._ptr(ws), nws if partial else 0, N, H, W, Cin, Cout, s, 0, ops._stream(dwp))
    dw = ops.conv_weight_unpermute(dwp, w.shape)
    if partial:  # same gradient written directly in the parameter's [Cout][Cin][3][3] layout
        dwt = torch.full((Cout, Cin, 3, 3), float("nan"), device=dev)
        ops.call("avsr_conv3x3_wgrad_bf16", ops._ptr(dyd), ops._ptr(xd), ops._ptr(dwt), ops._ptr(ops.zero_page(dev)),
                 ops._ptr(ws), nws, N, H, W, Cin, Cout, s, 1, ops._stream(dwt))
        assert torch.equal(dwt.cpu(), dw.cpu())
    assert (dw.cpu() - w.grad).abs().max() < 2e-3 * max(1.0, w.grad.abs().max().item())


@pytest.mark.parametrize("precise", [True, False])
def test_conv_stem_video(dev, precise):
    torch.manual_seed(3)
    B, T, H, W, Cout = 2, 5, 20, 18, 16
    x = torch.randn(B, T, H, W)
    w = (torch.randn(Cout, 1, 5, 7, 7) / 245 ** 0.5).requires_grad_()
    dtype = torch.float32 if precise else torch.bfloat16
    wq = w.detach().to(dtype).float().requires_grad_()
    xq = x.to(dtype).float() if not precise else x
    y_ref = F.conv3d(xq.unsqueeze(1), wq, stride=(1, 2, 2), padding=(2, 3, 3))  # (B,C,T,OH,OW)
    dy = torch.randn_like(y_ref).to(dtype).float()
    y_ref.backward(dy)
    tol = 3e-5 if precise else 3e-2
    wp = ops.conv_weight_permute(w.detach().to(dev), dtype, ld_out=248)
    y = ops.conv_stem_fwd(x.to(dev), wp, 248, dtype, B, T, H, W, Cout, 5, 7, 7, 2, 2, 3, 3, precise)
    ref = y_ref.detach().permute(0, 2, 3, 4, 1).reshape(B * T, y_ref.shape[3], y_ref.shape[4], Cout)
    assert (y.float().cpu() - ref).abs().max() < tol * max(1.0, ref.abs().max().item())
    dyd = dy.permute(0, 2, 3, 4, 1).reshape(ref.shape).contiguous().to(dtype).to(dev)
    dw = ops.conv_stem_wgrad(dyd, x.to(dev), B, T, H, W, Cout, 5, 7, 7, 2, 2, 3, 3, precise)
    assert (dw.cpu() - wq.grad.reshape(Cout, -1)).abs().max() < tol * max(1.0, wq.grad.abs().max().item())


def test_conv_stem_audio(dev):
    torch.manual_seed(4)
    B, S, Cout = 2, 640, 16
    x = torch.randn(B, S)
    w = (torch.randn(Cout, 1, 80) / 80 ** 0.5).requires_grad_()
    y_ref = F.conv1d(x.unsqueeze(1), w, stride=4, padding=38)  # (B, C, S/4)
    dy = torch.randn_like(y_ref)
    y_ref.backward(dy)
    wp = ops.conv_weight_permute(w.detach().to(dev), torch.float32, ld_out=80)
    y = ops.conv_stem_fwd(x.to(dev), wp, 80, torch.float32, B, 1, 1, S, Cout, 1, 1, 80, 4, 0, 0, 38, True)
    ref = y_ref.detach().permute(0, 2, 1).reshape(B, 1, S // 4, Cout)
    assert (y.cpu() - ref).abs().max() < 3e-5 * max(1.0, ref.abs().max().item())
    dyd = dy.permute(0, 2, 1).reshape(ref.shape).contiguous().to(dev)
    dw = ops.conv_stem_wgrad(dyd, x.to(dev), B, 1, 1, S, Cout, 1, 1, 80, 4, 0, 0, 38, True)
    assert (dw.cpu() - w.grad.reshape(Cout, -1)).abs().max() < 3e-5 * max(1.0, w.grad.abs().max().item())


def test_pooling(dev):
    torch.manual_seed(5)
    N, C, H, W = 3, 16, 11, 10
    x = torch.randn(N, C, H, W)
    x[0, :, 2, 2] = x[0, :, 2, 3]  # ties inside a window: first maximum wins
    x.requires_grad_()
    y_ref = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn_like(y_ref)
    y_ref.backward(dy)
    xd = nhwc(x.detach()).to(dev)
    y, idx = ops.maxpool2d_fwd(xd, N, H, W, C, 3, 2, 1)
    assert (y.cpu() - nhwc(y_ref.detach())).abs().max() == 0
    dx = ops.maxpool2d_bwd(idx, nhwc(dy).to(dev), N, H, W, C, 3, 2, 1)
    assert (dx.cpu() - nhwc(x.grad)).abs().max() < 1e-6
    z = torch.randn(4 * 9, 24)
    a = ops.avgpool_fwd(z.to(dev), 4, 9, 24)
    assert (a.cpu() - z.view(4, 9, 24).mean(1)).abs().max() < 1e-6
    dz = ops.avgpool_bwd(a, torch.float32, 4, 9, 24)
    assert (dz.cpu() - (a.cpu() / 9).repeat_interleave(9, 0)).abs().max() < 1e-6


def test_stem357_dedicated(dev):
    """Dedicated bf16 stem kernels (LDS-staged input rows) vs torch conv3d on bf16-rounded operands."""
    torch.manual_seed(6)
    B, T, H, W = 2, 4, 20, 24
    x = torch.randn(B, T, H, W)
    w = (torch.randn(64, 1, 5, 7, 7) / 245 ** 0.5)
    xq, wq = x.bfloat16().float(), w.bfloat16().float().requires_grad_()
    y_ref = F.conv3d(xq.unsqueeze(1), wq, stride=(1, 2, 2), padding=(2, 3, 3))
    dy = torch.randn_like(y_ref).bfloat16().float()
    y_ref.backward(dy)
    y = ops.stem357_fwd(x.to(dev), w.to(dev), B, T, H, W)
    ref = y_ref.detach().permute(0, 2, 3, 4, 1).reshape(B * T, y_ref.shape[3], y_ref.shape[4], 64)
    assert (y.float().cpu() - ref).abs().max() < 2e-2 * max(1.0, ref.abs().max().item())
    dyd = dy.permute(0, 2, 3, 4, 1).reshape(ref.shape).contiguous().bfloat16().to(dev)
    dw = ops.stem357_wgrad(dyd, x.to(dev), B, T, H, W)
    assert (dw.cpu() - wq.grad).abs().max() < 2e-2 * max(1.0, wq.grad.abs().max().item())


@pytest.mark.parametrize("geom", [(2, 4, 20, 24), (1, 3, 88, 88)])
def test_stem357_split_forward(dev, geom):
    """Precise-mode video stem on the dedicated kernel (split hi / lo planes, f32 result) vs torch conv3d in f64."""
    torch.manual_seed(7)
    B, T, H, W = geom
    x = torch.randn(B, T, H, W)
    w = torch.randn(64, 1, 5, 7, 7) / 245 ** 0.5
    y_ref = F.conv3d(x.double().unsqueeze(1), w.double(), stride=(1, 2, 2), padding=(2, 3, 3)).float()
    y = ops.stem357_fwd_f32s(x.to(dev), w.to(dev), B, T, H, W)
    ref = y_ref.permute(0, 2, 3, 4, 1).reshape(B * T, y_ref.shape[3], y_ref.shape[4], 64)
    assert y.dtype == torch.float32 and (y.cpu() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cfg", [(3, 22, 22), (2, 9, 10), (5, 1, 37), (2, 30, 12), (300, 7, 5), (1100, 6, 4)])
def test_conv3x3_c64_persistent(dev, cfg):
    """64 -> 64 channel 3x3 / stride-1 convolution on the weights-in-registers, patch-staged persistent kernel
    (conv3x3_c64.hip; forward and data gradient + residual) vs torch conv2d autograd on bf16-rounded operands and vs the
    tiled kernel it replaces (avsr_tune knob 12): several bands per image, ragged last band, more tiles than blocks."""
    N, H, W = cfg
    torch.manual_seed(H * 100 + W)
    C = 64
    x = torch.randn(N, C, H, W).bfloat16().float().requires_grad_()
    w = torch.randn(C, C, 3, 3) / (C * 9) ** 0.5
    wq = w.bfloat16().float()
    y_ref = F.conv2d(x, wq, stride=1, padding=1)
    dy = torch.randn_like(y_ref).bfloat16().float()
    y_ref.backward(dy)
    xd = nhwc(x.detach()).bfloat16().to(dev)
    dyd = nhwc(dy).bfloat16().to(dev)
    res = torch.randn(N, H, W, C).bfloat16()
    wp = ops.conv_weight_permute(w.to(dev), torch.bfloat16)
    wpd = ops.conv_weight_permute(w.to(dev), torch.bfloat16, to_dgrad=True)
    outs = []
    try:
        for knob in (0, 1):
            ops.tune(12, knob)
            y = ops.conv2d_fwd(xd, wp, N, H, W, C, C, 3, 3, 1, 1, 1, False)
            dx = ops.conv2d_dgrad(dyd, wpd, res.to(dev), N, H, W, C, C, 3, 3, 1, 1, 1, False)
            outs.append((y.float().cpu(), dx.float().cpu()))
    finally:
        ops.tune(12, 0)
    (y, dx), (yt, dxt) = outs
    assert (y - nhwc(y_ref.detach())).abs().max() < 2e-2 * max(1.0, y_ref.abs().max().item())
    assert (dx - nhwc(x.grad) - res.float()).abs().max() < 4e-2 * max(1.0, x.grad.abs().max().item())
    assert (y - yt).abs().max() < 2e-2 * max(1.0, y_ref.abs().max().item()) and (dx - dxt).abs().max() < 4e-2 * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("cfg", [(31, 3, 3, 128, 128), (9, 6, 6, 64, 256), (5, 11, 11, 128, 128), (30, 3, 3, 192, 384), (3, 5, 7, 64, 128)])
def test_conv_patch_staged(dev, cfg):
    """Round 6, conv_patch.hip: 3x3 / stride-1 convolutions on whole-image tiles with the input patch staged once per 64-channel
    chunk (forward, data gradient + residual in bf16; f16 forward with one and two weight planes) against torch conv2d on the
    same rounded operands and against the tiled kernel it replaces (knob 20): several tiles, a ragged last tile (fewer images
    than a tile holds), rows beyond the tile's images, two output-column tiles, odd image shapes."""
    from auto_avsr_amd import functional as AF

    N, H, W, Cin, Cout = cfg
    torch.manual_seed(N * 10 + H)
    x = torch.randn(N, Cin, H, W).bfloat16().float().requires_grad_()
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    wq = w.bfloat16().float()
    y_ref = F.conv2d(x, wq, stride=1, padding=1)
    dy = torch.randn_like(y_ref).bfloat16().float()
    y_ref.backward(dy)
    xd = nhwc(x.detach()).bfloat16().to(dev)
    dyd = nhwc(dy).bfloat16().to(dev)
    res = torch.randn(N, H, W, Cin).bfloat16()
    wp = ops.conv_weight_permute(w.to(dev), torch.bfloat16)
    wpd = ops.conv_weight_permute(w.to(dev), torch.bfloat16, to_dgrad=True)
    outs = []
    try:
        for knob in (2, 1, 3):  # 2 / 3 = the patch-staged kernel (two / three weight stages) whatever the grid size, 1 = the tiled kernel
            ops.tune(20, knob)
            y = ops.conv2d_fwd(xd, wp, N, H, W, Cin, Cout, 3, 3, 1, 1, 1, False)
            dx = ops.conv2d_dgrad(dyd, wpd, res.to(dev), N, H, W, Cin, Cout, 3, 3, 1, 1, 1, False) if Cin % 128 == 0 else None
            outs.append((y.float().cpu(), dx.float().cpu() if dx is not None else None))
    finally:
        ops.tune(20, 0)
    (y, dx), (yt, dxt), (y3, dx3) = outs
    assert torch.equal(y3, y) and (dx is None or torch.equal(dx3, dx))  # the ring depth changes no arithmetic
    assert (y - nhwc(y_ref.detach())).abs().max() < 2e-2 * max(1.0, y_ref.abs().max().item())
    assert (y - yt).abs().max() < 2e-2 * max(1.0, y_ref.abs().max().item())
    if dx is not None:
        assert (dx - nhwc(x.grad) - res.float()).abs().max() < 4e-2 * max(1.0, x.grad.abs().max().item())
        assert (dx - dxt).abs().max() < 4e-2 * max(1.0, x.grad.abs().max().item())
    # f16 forward, one and two weight planes (the mixed mode's trunk stages 3 - 4)
    AF.invalidate_weight_cache()
    xh = nhwc(x.detach()).half()
    buf = AF._w_conv_h16(w.to(dev))
    ref2 = F.conv2d(xh.double().permute(0, 3, 1, 2), w.double(), stride=1, padding=1).permute(0, 2, 3, 1)
    ref1 = F.conv2d(xh.double().permute(0, 3, 1, 2), w.half().double(), stride=1, padding=1).permute(0, 2, 3, 1)
    try:
        ops.tune(20, 2)
        y2 = ops.conv2d_fwd(xh.to(dev), buf, N, H, W, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=2)
        y1 = ops.conv2d_fwd(xh.to(dev), buf, N, H, W, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=1)
    finally:
        ops.tune(20, 0)
        AF.invalidate_weight_cache()

    def rel(a, b):
        return float((a.double().cpu() - b).norm() / b.norm())

    assert y2.dtype == torch.float16 and rel(y2, ref2) < 2.5e-4 and rel(y1, ref1) < 2.5e-4  # (the f16 rounding of the output)


def test_multi_weight_permute_matches_single(dev):
    """All conv-weight copies of a model in ONE launch (LDS-transposed tiles: 8 co x 64 ci forward copies, 64 co x 8 ci
    data-gradient copies, all taps) == the per-tensor permute, bit for bit; ragged channel counts and 1 / 3 / 9 taps included."""
    import struct

    torch.manual_seed(11)
    shapes = [(64, 64, 9, 0), (64, 64, 9, 1), (128, 64, 9, 0), (128, 64, 9, 1), (128, 64, 1, 0), (128, 64, 1, 1), (72, 40, 3, 0),
              (72, 40, 3, 1), (8, 200, 9, 1), (200, 8, 9, 0)]
    ws, outs, blob, blk = [], [], b"", 0
    for (Cout, Cin, taps, dg) in shapes:
        w = torch.randn(Cout, Cin, taps, device=dev)
        o = torch.full((Cout * Cin * taps,), float("nan"), dtype=torch.bfloat16, device=dev)
        blob += struct.pack("<QQiiiiiiii", w.data_ptr(), o.data_ptr(), Cout, Cin, taps, dg, blk, 0, 0, 0)
        blk += ops.weight_permute_blocks(Cout, Cin, dg)
        ws.append(w)
        outs.append(o)
    table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    ops.multi_weight_permute(table, len(shapes), blk, 9)
    for (Cout, Cin, taps, dg), w, o in zip(shapes, ws, outs):
        ref = ops.conv_weight_permute(w.view(Cout, Cin, taps, 1), torch.bfloat16, to_dgrad=bool(dg))
        assert torch.equal(o.cpu().view(-1), ref.cpu().view(-1)), (Cout, Cin, taps, dg)


def test_multi_weight_permute_split8_matches_single(dev):
    """Round 6: the split8 (hi + lo bf16 planes) permuted copies of the split-plane forward convolutions leave the same table
    launch (entry code 3) -- bit for bit the per-tensor avsr_conv_weight_permute(out_dtype = 2) they replace."""
    import struct

    torch.manual_seed(12)
    shapes = [(64, 64, 9), (128, 64, 9), (128, 64, 1), (128, 128, 3), (72, 64, 9)]
    ws, outs, blob, blk = [], [], b"", 0
    for (Cout, Cin, taps) in shapes:
        w = torch.randn(Cout, Cin, taps, 1, device=dev)
        o = ops.conv_weight_permute_split(torch.zeros_like(w))
        blob += struct.pack("<QQiiiiiiii", w.data_ptr(), o.data_ptr(), Cout, Cin, taps, 0, blk, 3, 0, 0)
        blk += ops.weight_permute_blocks(Cout, Cin, False)
        ws.append(w)
        outs.append(o)
    table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    ops.multi_weight_permute(table, len(shapes), blk, 9)
    for w, o in zip(ws, outs):
        ref = ops.conv_weight_permute_split(w)
        assert torch.equal(o.as_subclass(torch.Tensor).cpu().view(torch.int32), ref.as_subclass(torch.Tensor).cpu().view(torch.int32)), w.shape
