"""Module-level parity of the HIP-backed nn.Module tree against (a) the reference-generated golden vectors and
(b) the oracle, in precise mode (split-bf16 contractions) with dropout off.  Runs on the emulator build in the
CPU suite and on the gfx950 build under -m gpu."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
from synth import synth_batch, synth_state_dict  # noqa: E402

import avsr_oracle as O  # noqa: E402
from auto_avsr_amd import functional as AF  # noqa: E402
from auto_avsr_amd import nets  # noqa: E402
from auto_avsr_amd.e2e import E2E  # noqa: E402


@pytest.fixture(scope="module")
def golden():
    return torch.load(os.path.join(HERE, "golden", "golden_v1.pt"), weights_only=False)


def no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-12)).item()


def check_grad_norms(model, ref_norms, rtol):
    atol = 1e-4 * max(ref_norms.values())
    bad = []
    for k, p in model.named_parameters():
        got = float(p.grad.double().norm()) if p.grad is not None else 0.0
        if abs(got - ref_norms[k]) > rtol * ref_norms[k] + atol:
            bad.append((k, got, ref_norms[k]))
    assert not bad, bad[:6]


def test_encoder_small_vs_reference_golden(dev, golden):
    c = golden["encoder_small"]
    enc = no_dropout(nets.ConformerEncoder(attention_dim=128, attention_heads=2, linear_units=256, num_blocks=2,
                                           cnn_module_kernel=7))
    assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == c["shapes"], "state_dict contract"
    enc.load_state_dict(synth_state_dict(enc.state_dict(), c["seed"]), strict=True)
    enc.to(dev).train()
    x = c["x"].clone().to(dev).requires_grad_()
    mask = nets.make_non_pad_mask(c["lengths"]).unsqueeze(-2).to(dev)
    with AF.precise():
        out, _ = enc(x, mask)
        assert rel(out.cpu(), c["out"]) < 1e-3
        (out * c["w"].to(dev)).sum().backward()
    assert rel(x.grad.cpu(), c["dx"]) < 1e-3
    check_grad_norms(enc, c["grad_norms"], 5e-3)


def test_decoder_small_vs_reference_golden(dev, golden):
    c = golden["decoder_small"]
    dec = no_dropout(nets.TransformerDecoder(odim=c["odim"], attention_dim=128, attention_heads=2, linear_units=256,
                                             num_blocks=2))
    assert {k: tuple(v.shape) for k, v in dec.state_dict().items()} == c["shapes"], "state_dict contract"
    dec.load_state_dict(synth_state_dict(dec.state_dict(), c["seed"]), strict=True)
    dec.to(dev).train()
    mem = c["memory"].clone().to(dev).requires_grad_()
    mmask = nets.make_non_pad_mask(c["lengths"]).unsqueeze(-2).to(dev)
    ys_in = c["ys_in"].to(dev)
    with AF.precise():
        out, _ = dec(ys_in, nets.target_mask(ys_in, -1), mem, mmask)
        assert rel(out.cpu(), c["out"]) < 1e-3
        (out * c["w"].to(dev)).sum().backward()
    assert rel(mem.grad.cpu(), c["dmemory"]) < 1e-3
    check_grad_norms(dec, c["grad_norms"], 5e-3)


@pytest.mark.parametrize("modality", ["video", "audio"])
def test_e2e_small_vs_oracle(dev, modality):
    torch.manual_seed(0)
    odim = 40
    m = no_dropout(E2E(odim, modality, adim=128, aheads=2, eunits=256, elayers=2, dunits=256, dlayers=2,
                       cnn_module_kernel=7))
    sd = synth_state_dict(m.state_dict(), 11)
    m.load_state_dict(sd, strict=True)
    m.to(dev).train()
    x, lengths, y = synth_batch(modality, 2, 9, 4, odim, seed=5)
    osd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running_" not in k else v.clone())
           for k, v in sd.items()}
    (loss_r, ctc_r, att_r, acc_r), _ = O.e2e_forward(osd, x, lengths, y, modality=modality, heads=2)
    loss_r.backward()
    with AF.precise():
        loss, loss_ctc, loss_att, acc = m(x.to(dev), lengths.to(dev), y.to(dev))
        loss.backward()
    assert abs(float(loss_ctc) - float(ctc_r)) < 1e-3 * abs(float(ctc_r))
    assert abs(float(loss_att) - float(att_r)) < 1e-3 * abs(float(att_r))
    assert abs(float(loss) - float(loss_r)) < 1e-3 * abs(float(loss_r))
    assert acc == acc_r
    ref_norms = {k: float(v.grad.double().norm()) for k, v in osd.items() if v.is_floating_point() and v.grad is not None}
    check_grad_norms(m, ref_norms, 1e-2)
    # BatchNorm running statistics of the conv module were updated like torch's (momentum 0.1, unbiased variance)
    assert int(m.encoder.encoders[0].conv_module.norm.num_batches_tracked) == 1
    # ... and so was every batch counter (incremented on the device by the statistics kernel)
    bns = [mod for mod in m.modules() if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm)]
    assert len(bns) > 3 and all(int(b.num_batches_tracked) == 1 for b in bns)


@pytest.mark.parametrize("modality", ["video"])
def test_e2e_small_bf16_mode(dev, modality):
    """bf16 bench mode (LDS-DMA NT kernel, bf16 weight / transposed activation copies): losses within 2e-2 of the
    fp32 oracle and gradients aligned with it."""
    torch.manual_seed(0)
    odim = 72
    m = no_dropout(E2E(odim, modality, adim=128, aheads=2, eunits=256, elayers=2, dunits=256, dlayers=2,
                       cnn_module_kernel=7))
    sd = synth_state_dict(m.state_dict(), 13)
    m.load_state_dict(sd, strict=True)
    m.to(dev).train()
    x, lengths, y = synth_batch(modality, 2, 9, 4, odim, seed=8)
    osd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running_" not in k else v.clone())
           for k, v in sd.items()}
    (loss_r, ctc_r, att_r, _), _ = O.e2e_forward(osd, x, lengths, y, modality=modality, heads=2)
    loss_r.backward()
    AF.invalidate_weight_cache()
    loss, loss_ctc, loss_att, acc = m(x.to(dev), lengths.to(dev), y.to(dev))
    loss.backward()
    assert abs(float(loss_ctc) - float(ctc_r)) < 2e-2 * abs(float(ctc_r))
    assert abs(float(loss_att) - float(att_r)) < 2e-2 * abs(float(att_r))
    cos = []
    for k, p in m.named_parameters():
        a, b = p.grad.double().flatten().cpu(), osd[k].grad.double().flatten()
        if b.norm() > 1e-4 * max(1.0, float(osd[k].double().norm())):
            cos.append((float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), k))
    assert min(cos)[0] > 0.9, sorted(cos)[:5]


@pytest.mark.parametrize("modality,twins", [("video", False), ("video", True), ("audio", True)], ids=["video-cast", "video-twins", "audio-twins"])
def test_e2e_small_hpf_mode(dev, modality, twins, monkeypatch):
    """"hpf" numerical mode (functional.set_mode): the FORWARD pass is the precise one -- losses bit-identical to the precise
    mode and within 1e-3 of the fp32 oracle (the north-star bound) -- while the backward pass runs the bf16 kernels on bf16
    copies of the saved activations: gradients aligned with the oracle's like the bf16 mode's."""
    torch.manual_seed(0)
    odim = 72
    m = no_dropout(E2E(odim, modality, adim=128, aheads=2, eunits=256, elayers=2, dunits=256, dlayers=2,
                       cnn_module_kernel=7))
    sd = synth_state_dict(m.state_dict(), 13)
    m.load_state_dict(sd, strict=True)
    m.to(dev).train()
    x, lengths, y = synth_batch(modality, 2, 9, 4, odim, seed=8)
    osd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running_" not in k else v.clone())
           for k, v in sd.items()}
    (loss_r, ctc_r, att_r, acc_r), _ = O.e2e_forward(osd, x, lengths, y, modality=modality, heads=2)
    loss_r.backward()
    AF.invalidate_weight_cache()
    with AF.precise():
        ref = [float(v) for v in m(x.to(dev), lengths.to(dev), y.to(dev))[:3]]
    m.load_state_dict(sd, strict=True)  # (running statistics back to the start)
    # twins: the producing kernels write the bf16 copies the backward pass reads (the size threshold is for real workloads);
    # cast: every saved activation is cast at save time -- the two must agree bit for bit
    monkeypatch.setattr(AF, "_TWIN_MIN", 0 if twins else 1 << 60)
    AF._twin_stats.update(made=0, used=0)
    with AF.numerics("hpf"):
        assert AF.mode() == "hpf"
        loss, loss_ctc, loss_att, acc = m(x.to(dev), lengths.to(dev), y.to(dev))
        loss.backward()
    if twins:
        assert AF._twin_stats["used"] > 20 and AF._twin_stats["used"] >= 0.9 * AF._twin_stats["made"] - 2, AF._twin_stats
    else:
        assert AF._twin_stats["made"] == 0
    gsum = sum(float(p.grad.double().abs().sum()) for p in m.parameters())
    key = ("hpf_gsum", modality)
    if key in _HPF_REF:
        # (equal up to the summation order of the float atomics in the weight-gradient / bias-gradient kernels)
        assert abs(gsum - _HPF_REF[key]) <= 1e-9 * gsum, "twin outputs and save-time casts must give the same gradients"
    _HPF_REF[key] = gsum
    assert AF.mode() == "bf16"
    assert [float(loss), float(loss_ctc), float(loss_att)] == ref, "hpf forward must be the precise forward"
    assert abs(float(loss_ctc) - float(ctc_r)) < 1e-3 * abs(float(ctc_r))
    assert abs(float(loss_att) - float(att_r)) < 1e-3 * abs(float(att_r))
    assert acc == acc_r
    cos = []
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32, k
        a, b = p.grad.double().flatten().cpu(), osd[k].grad.double().flatten()
        if b.norm() > 1e-4 * max(1.0, float(osd[k].double().norm())):
            cos.append((float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)), k))
    assert min(cos)[0] > 0.9, sorted(cos)[:5]
    AF.invalidate_weight_cache()


_HPF_REF = {}


def test_weight_cache_refresh(dev):
    """bf16 weight copies: lazy build, in-place refresh after an optimizer-style update, multi-tensor refresh."""
    torch.manual_seed(3)
    AF.invalidate_weight_cache()
    lin1, lin2 = torch.nn.Linear(64, 72).to(dev), torch.nn.Linear(128, 200).to(dev)
    x1, x2 = torch.randn(5, 64, device=dev).bfloat16(), torch.randn(7, 128, device=dev).bfloat16()

    def run():
        y1 = AF.linear(x1, lin1.weight, lin1.bias, out_dtype=torch.float32)
        y2 = AF.linear(x2.requires_grad_(), lin2.weight, lin2.bias, out_dtype=torch.float32)
        y2.sum().backward()  # registers the transposed copy of lin2.weight (data gradient)
        return y1, y2

    y1, y2 = run()
    with torch.no_grad():
        lin1.weight.mul_(2.0)
        lin2.weight.add_(1.0)
    AF.refresh_weight_cache()  # one launch for all registered copies
    x2.grad = None
    z1, z2 = run()
    ref1 = x1.float().cpu() @ lin1.weight.detach().cpu().bfloat16().float().t() + lin1.bias.detach().cpu()
    ref2 = x2.detach().float().cpu() @ lin2.weight.detach().cpu().bfloat16().float().t() + lin2.bias.detach().cpu()
    assert (z1.cpu() - ref1).abs().max() < 1e-3 and (z2.detach().cpu() - ref2).abs().max() < 1e-3
    gx = lin2.weight.detach().cpu().bfloat16().float().sum(0)
    assert (x2.grad.float().cpu() - gx).abs().max() < 0.02 * gx.abs().max()
    AF.invalidate_weight_cache()


@pytest.mark.parametrize("relpos", [True, False])
def test_fused_qkv_projection(dev, relpos):
    """bf16 self attention runs ONE projection GEMM onto [Wq; Wk; Wv] (and one weight- / data-gradient GEMM in the
    backward).  Must agree with the three-GEMM path (forced through precise mode's f32 reference within bf16
    tolerance), keep agreeing after an optimizer-style weight update + refresh_weight_cache(), and hand back
    per-parameter gradients."""
    torch.manual_seed(11)
    AF.invalidate_weight_cache()
    B, T, D, H = 2, 9, 128, 2
    from auto_avsr_amd import nets
    att = (nets.RelPositionMultiHeadedAttention(H, D, 0.0) if relpos else nets.MultiHeadedAttention(H, D, 0.0)).to(dev)
    ln = torch.nn.LayerNorm(D).to(dev)
    x = torch.randn(B, T, D, device=dev)
    pos = torch.randn(1, 2 * T - 1, D, device=dev) if relpos else None
    mask = torch.ones(B, 1, T, dtype=torch.bool, device=dev)
    mask[1, 0, 6:] = False
    params = [att.linear_q, att.linear_k, att.linear_v, att.linear_out]

    def run(precise):
        AF.set_precise(precise)
        for p in att.parameters():
            p.grad = None
        xx = x.clone().requires_grad_()
        extra = (att.linear_pos.weight, att.pos_bias_u, att.pos_bias_v) if relpos else (None, None, None)
        y = AF.mha_sublayer(xx, None, pos, mask, ln.weight, ln.bias, att.linear_q.weight, att.linear_q.bias,
                            att.linear_k.weight, att.linear_k.bias, att.linear_v.weight, att.linear_v.bias,
                            att.linear_out.weight, att.linear_out.bias, *extra, H, 0.0, 0.0)
        (y * torch.linspace(-1, 1, D, device=dev)).sum().backward()
        return y.detach().float().cpu(), xx.grad.float().cpu(), [l.weight.grad.float().cpu().clone() for l in params], \
            [l.bias.grad.float().cpu().clone() for l in params]

    try:
        for round_ in range(2):
            ref = run(True)
            got = run(False)
            assert (got[0] - ref[0]).abs().max() < 0.05 * ref[0].abs().max()
            assert (got[1] - ref[1]).abs().max() < 0.05 * ref[1].abs().max()
            for grp in (2, 3):  # weights, biases; dL/d(bias_k) is analytically zero (softmax shift invariance), so
                floor = max(b.abs().max().item() for b in ref[grp])  # tolerances are relative to the group's scale
                for a, b in zip(got[grp], ref[grp]):
                    assert a.shape == b.shape and (a - b).abs().max() < 0.05 * max(b.abs().max().item(), 0.2 * floor)
            with torch.no_grad():  # optimizer step: every cached bf16 copy (concatenated or not) is stale now
                for l in params:
                    l.weight.mul_(0.5)
                    l.weight.add_(0.01)
            AF.refresh_weight_cache()
    finally:
        AF.set_precise(False)
        AF.invalidate_weight_cache()


def test_linear_padded_head_f32_grad_bf16_mode(dev):
    """Vocabulary-style head in the bf16 mode: y = x W^T + b with an odd width (121 -> row pitch 128), f32 output, f32
    output gradient.  The backward casts the pitched gradient once (bias gradient in the same pass) and runs the data
    and weight gradients on the bf16 kernels as one paired launch; compared with torch on the bf16-rounded operands."""
    torch.manual_seed(21)
    rows, K, N = 96, 64, 121
    was_precise = AF._state["precise"]
    AF.set_precise(False)
    AF.invalidate_weight_cache()
    try:
        x = (torch.randn(2, rows // 2, K) * 0.5).bfloat16()
        w = torch.nn.Parameter(torch.randn(N, K) * 0.2)
        b = torch.nn.Parameter(torch.randn(N) * 0.1)
        xd = x.clone().to(dev).requires_grad_()
        wd, bd = torch.nn.Parameter(w.detach().to(dev)), torch.nn.Parameter(b.detach().to(dev))
        y = AF.linear(xd, wd, bd, out_dtype=torch.float32, pad_out=True)
        assert y.shape == (2, rows // 2, N) and y.stride(-2) == 128
        gy = torch.randn(2, rows // 2, N)
        (y * gy.to(dev)).sum().backward()
        xr = x.float().requires_grad_()
        wr = w.detach().bfloat16().float().requires_grad_()
        yr = xr @ wr.t() + b
        (yr * gy).sum().backward()
        assert rel(y.detach().cpu(), yr.detach()) < 1e-5
        gyb = gy.bfloat16().float()  # the backward rounds the output gradient to bf16
        assert rel(wd.grad.cpu(), gyb.reshape(rows, N).t() @ x.float().reshape(rows, K)) < 1e-5
        assert rel(bd.grad.cpu(), gyb.reshape(rows, N).sum(0)) < 1e-5
        assert rel(xd.grad.float().cpu(), (gyb.reshape(rows, N) @ wr.detach()).reshape(2, rows // 2, K)) < 1e-2
    finally:
        AF.invalidate_weight_cache()
        AF.set_precise(was_precise)


def test_stem_fused_pool_switch_equivalence(dev):
    """AVSR_FUSE_STEM_POOL (BN + SiLU + max-pool of the video stem in one pass): same losses and gradients as the two-pass
    path, bf16 mode."""
    was_precise, was_fused = AF._state["precise"], AF._FUSE_STEM_POOL
    odim = 40
    res = []
    try:
        AF.set_precise(False)
        for fused in (False, True):
            AF._FUSE_STEM_POOL = fused
            AF.invalidate_weight_cache()
            torch.manual_seed(0)
            m = no_dropout(E2E(odim, "video", adim=64, aheads=1, eunits=64, elayers=1, dunits=64, dlayers=1, cnn_module_kernel=7))
            m.load_state_dict(synth_state_dict(m.state_dict(), 3), strict=True)
            m.to(dev).train()
            x, lengths, y = (t.to(dev) for t in synth_batch("video", 2, 7, 3, odim, seed=2))
            loss, loss_ctc, loss_att, _ = m(x, lengths, y)
            loss.backward()
            res.append((float(loss_ctc), float(loss_att), {k: p.grad.cpu().clone() for k, p in m.named_parameters()}))
    finally:
        AF._FUSE_STEM_POOL = was_fused
        AF.set_precise(was_precise)
        AF.invalidate_weight_cache()
    (c0, a0, g0), (c1, a1, g1) = res
    assert abs(c0 - c1) < 1e-4 * abs(c0) and abs(a0 - a1) < 1e-4 * abs(a0)
    for k in g0:
        # bf16 mode: the fused path skips one bf16 rounding of the stem's activation gradient, and where several taps of a
        # pooling window round to the same bf16 activation it may route the gradient to another of the tied taps (round 4: the
        # maximum comes from the window's largest / smallest input, two activations instead of nine)
        assert rel(g1[k], g0[k]) < 2e-2 or float(g0[k].abs().max()) < 1e-7, k


@pytest.mark.parametrize("mode", ["bf16", "mixed"])
def test_convmod_fused_middle_switch_equivalence(dev, mode):
    """AVSR_CONVMOD_FUSED (round 6, opt-in: GLU -> depthwise conv -> BatchNorm -> SiLU of the convolution module and its backward as
    one launch each): the same losses, gradients and BatchNorm running statistics as the launches it merges -- the kernels are
    bit-equal but for the summation order of the depthwise weight gradient."""
    was_mode, was_fused = AF.mode(), AF._CONVMOD_FUSED
    odim = 40
    res = []
    try:
        AF.set_mode(mode)
        for fused in (False, True):
            AF._CONVMOD_FUSED = fused
            AF.invalidate_weight_cache()
            torch.manual_seed(0)
            m = no_dropout(E2E(odim, "audio", adim=64, aheads=1, eunits=64, elayers=2, dunits=64, dlayers=1, cnn_module_kernel=7))
            m.load_state_dict(synth_state_dict(m.state_dict(), 3), strict=True)
            m.to(dev).train()
            x, lengths, y = (t.to(dev) for t in synth_batch("audio", 2, 7, 3, odim, seed=2))
            loss, loss_ctc, loss_att, _ = m(x, lengths, y)
            loss.backward()
            bn = m.encoder.encoders[0].conv_module.norm
            res.append((float(loss_ctc), float(loss_att), {k: p.grad.cpu().clone() for k, p in m.named_parameters()},
                        bn.running_mean.cpu().clone(), bn.running_var.cpu().clone(), int(bn.num_batches_tracked)))
    finally:
        AF._CONVMOD_FUSED = was_fused
        AF.set_mode(was_mode)
        AF.invalidate_weight_cache()
    (c0, a0, g0, rm0, rv0, n0), (c1, a1, g1, rm1, rv1, n1) = res
    assert c0 == c1 and a0 == a1 and n0 == n1 == 1
    assert torch.equal(rm0, rm1) and torch.equal(rv0, rv1)
    for k in g0:
        assert rel(g1[k], g0[k]) < 1e-4 or float(g0[k].abs().max()) < 1e-7, k


def test_residual_gradient_handoff_equivalence(dev):
    """The backward prologue of a sub-layer's output Linear produced inside the NEXT sub-layer's LayerNorm backward
    (functional._chain_*) vs the stand-alone cast / column-sum launch: same gradients (dropout ON -- both paths must draw
    the same masks), and the stand-alone launch really is gone from the encoder / decoder sub-layer chains."""
    from auto_avsr_amd import ops

    was_precise, was_chain = AF._state["precise"], AF._CHAIN
    odim = 40
    res, counts = [], []
    orig = ops.cast_transpose_colsum
    try:
        AF.set_precise(False)
        for chain in (False, True):
            AF._CHAIN = chain
            AF.invalidate_weight_cache()
            AF.new_step()
            AF.manual_seed(77)
            torch.manual_seed(0)
            m = E2E(odim, "video", adim=128, aheads=2, eunits=128, elayers=2, dunits=128, dlayers=2, cnn_module_kernel=7)
            m.load_state_dict(synth_state_dict(m.state_dict(), 5), strict=True)
            m.to(dev).train()  # dropout 0.1 active
            x, lengths, y = (t.to(dev) for t in synth_batch("video", 2, 9, 3, odim, seed=4))
            n = [0]

            def counting(src, R, C, **kw):
                if kw.get("want_dst") and src.dtype == torch.float32:
                    n[0] += 1
                return orig(src, R, C, **kw)

            ops.cast_transpose_colsum = counting
            loss, loss_ctc, loss_att, _ = m(x, lengths, y)
            loss.backward()
            ops.cast_transpose_colsum = orig
            counts.append(n[0])
            res.append((float(loss.detach()), {k: p.grad.float().cpu().clone() for k, p in m.named_parameters()}))
    finally:
        ops.cast_transpose_colsum = orig
        AF._CHAIN = was_chain
        AF.set_precise(was_precise)
        AF.invalidate_weight_cache()
        AF.new_step()
    (l0, g0), (l1, g1) = res
    assert l0 == l1
    # 2 encoder layers x 4 sub-layers + 2 decoder layers x 3 = 14 f32 prologues without the hand-off; with it only the
    # first sub-layer after a non-LayerNorm consumer... none: every sub-layer output feeds a LayerNorm
    assert counts[0] - counts[1] >= 12, counts
    for k in g0:
        assert rel(g1[k], g0[k]) < 1e-5 or float(g0[k].abs().max()) < 1e-7, k


def test_decoder_shared_memory_projection_equivalence(dev):
    """Source-attention K / V of all decoder layers from ONE projection of the memory (functional.MemoryKVFn) vs the
    per-layer projections: same loss and gradients (bf16 mode, dropout on), and 2 * dlayers - 1 fewer forward GEMMs."""
    from auto_avsr_amd import ops

    was_precise, was_fuse = AF._state["precise"], AF._FUSE_QKV
    odim = 40
    res, counts = [], []
    orig = ops.gemm_bf16_nt
    try:
        AF.set_precise(False)
        for shared in (False, True):
            AF.invalidate_weight_cache()
            AF.new_step()
            AF.manual_seed(78)
            torch.manual_seed(0)
            m = E2E(odim, "audio", adim=128, aheads=2, eunits=128, elayers=1, dunits=128, dlayers=3, cnn_module_kernel=7)
            m.load_state_dict(synth_state_dict(m.state_dict(), 6), strict=True)
            m.to(dev).train()
            x, lengths, y = (t.to(dev) for t in synth_batch("audio", 2, 9, 3, odim, seed=5))
            n = [0]

            def counting(*a, **kw):
                n[0] += 1
                return orig(*a, **kw)

            real = AF.memory_kv
            if not shared:
                AF.memory_kv = lambda *a, **k: None
            ops.gemm_bf16_nt = counting
            try:
                loss, _, _, _ = m(x, lengths, y)
                fwd = n[0]
                loss.backward()
            finally:
                ops.gemm_bf16_nt = orig
                AF.memory_kv = real
            counts.append(fwd)
            res.append((float(loss.detach()), {k: p.grad.float().cpu().clone() for k, p in m.named_parameters()}))
    finally:
        AF._FUSE_QKV = was_fuse
        AF.set_precise(was_precise)
        AF.invalidate_weight_cache()
        AF.new_step()
    (l0, g0), (l1, g1) = res
    assert abs(l0 - l1) < 1e-6 * abs(l0), (l0, l1)
    assert counts[0] - counts[1] == 2 * 3 - 1, counts
    for k in g0:
        if k.endswith("linear_k.bias"):
            continue  # zero in exact arithmetic (softmax is invariant to a shift of every score of a row): pure rounding noise
        # the memory gradient is one K = 2*n*D contraction instead of a chain of bf16-rounded partial sums: equal to bf16 noise
        assert rel(g1[k], g0[k]) < 2e-2 or float(g0[k].abs().max()) < 1e-7, (k, rel(g1[k], g0[k]))
    for k in g0:
        if "decoder" in k and "src_attn" in k and not k.endswith("linear_k.bias"):
            assert rel(g1[k], g0[k]) < 2e-3, (k, rel(g1[k], g0[k]))


def test_encoder_shared_position_projection_equivalence(dev):
    """linear_pos of all encoder layers as one autograd node (functional.PosProjFn: one forward GEMM, one weight-gradient
    contraction over a shared [P, n*D] f32 buffer) vs per-layer projections: identical loss, gradients to rounding."""
    was_precise = AF._state["precise"]
    odim = 40
    res = []
    real = AF.prepare_pos_proj
    try:
        AF.set_precise(False)
        for shared in (False, True):
            AF.invalidate_weight_cache()
            AF.new_step()
            AF.manual_seed(79)
            torch.manual_seed(0)
            m = E2E(odim, "audio", adim=128, aheads=2, eunits=128, elayers=3, dunits=128, dlayers=1, cnn_module_kernel=7)
            m.load_state_dict(synth_state_dict(m.state_dict(), 7), strict=True)
            m.to(dev).train()
            x, lengths, y = (t.to(dev) for t in synth_batch("audio", 2, 9, 3, odim, seed=6))
            AF.prepare_pos_proj = real if shared else (lambda *a, **k: AF._pos_proj.clear())
            try:
                loss, _, _, _ = m(x, lengths, y)
                loss.backward()
            finally:
                AF.prepare_pos_proj = real
            res.append((float(loss.detach()), {k: p.grad.float().cpu().clone() for k, p in m.named_parameters()}))
    finally:
        AF.prepare_pos_proj = real
        AF.set_precise(was_precise)
        AF.invalidate_weight_cache()
        AF.new_step()
    (l0, g0), (l1, g1) = res
    assert l0 == l1
    seen = 0
    for k in g0:
        tol = 1e-5
        assert rel(g1[k], g0[k]) < tol or float(g0[k].abs().max()) < 1e-7, (k, rel(g1[k], g0[k]))
        seen += "linear_pos" in k
    assert seen == 3


def test_downsampling_blocks_share_their_batchnorm_all_gather(dev):
    """Cross-rank BatchNorm (train.py:31): in a down-sampling residual block the statistics of the main path's first BatchNorm
    and of the down-sampling path's cross the ranks in ONE all-gather (round 5).  A one-rank stand-in communicator counts the
    collectives of a training-mode forward + backward pass: one all-gather per BatchNorm minus one per down-sampling block (3),
    one all-reduce per BatchNorm in the backward pass; losses equal the unsynchronised run."""
    from synth import synth_batch, synth_state_dict

    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.e2e import E2E

    class Comm:
        world, rank = 1, 0

        def __init__(self):
            self.gathers, self.reduces = [], 0

        def all_gather(self, out, mine):
            self.gathers.append(mine.numel())
            out.copy_(mine.reshape(-1))

        def all_reduce(self, t):
            self.reduces += 1
            return t

    torch.manual_seed(0)
    AF.invalidate_weight_cache()
    m = no_dropout(E2E(40, "video", adim=128, aheads=2, eunits=64, elayers=1, dunits=64, dlayers=1, cnn_module_kernel=7))
    sd = synth_state_dict(m.state_dict(), 5)
    x, lens, y = synth_batch("video", 2, 7, 3, 40, seed=2)
    res = {}
    for sync in (False, True):
        m.load_state_dict(sd)
        m.to(dev).train()
        comm = Comm()
        AF.set_bn_sync("stand-in group" if sync else None, comm=comm if sync else None)
        try:
            with AF.precise():
                loss, *_ = m.forward_tensors(x.to(dev), lens.to(dev), y.to(dev))
                loss.backward()
        finally:
            AF.set_bn_sync(None)
        res[sync] = (float(loss), comm)
        m.zero_grad(set_to_none=True)
    n_bn = sum(isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) for mod in m.modules())
    comm = res[True][1]
    assert n_bn == 1 + 16 + 3 + 1  # stem, 8 blocks x 2, 3 down-sampling paths, the convolution module's
    assert len(comm.gathers) == n_bn - 3 and comm.reduces == n_bn, (len(comm.gathers), comm.reduces, n_bn)
    assert sum(1 for n in comm.gathers if n in (2 * (3 * 128 + 1), 2 * (3 * 256 + 1), 2 * (3 * 512 + 1))) == 3  # the three paired payloads
    assert abs(res[True][0] - res[False][0]) < 1e-5 * abs(res[False][0])
    AF.invalidate_weight_cache()
