"""Autograd glue of the front-ends (split out of functional.py in round 4): BasicBlockFn (ResNet block), StemFn (Conv3d /
Conv1d stem + BatchNorm + SiLU (+ max-pool)), AvgPoolFn.  Re-exported by functional.py."""

import os

import torch

from . import functional as AF
from . import ops
from .functional import (  # noqa: F401
    _A, _act_in, _bwd_mode, _bwd_precise, _f32_in, _hand_over, _state, _to_act, _to_f32, _twins, _w_conv,
    _w_conv_fwd, act_dtype, padded_cols)
from .functional_convmod import _bn_train_stats_pair  # noqa: F401
from .functional_convmod import (  # noqa: F401
    _bn_bwd_sums, _bn_train_stats)


# ================================================================================================ front-ends
def _bn_fwd_params(c2, rows, C, bn, training, parts=None):
    """(mean, invstd, counts) of a BatchNorm over the rows of c2; bn = (weight, bias, running_mean, running_var,
    eps, momentum).  Training: batch statistics (cross-rank when set_bn_sync) + running-stat update; parts: partial statistics
    from the producing convolution's epilogue (_conv_bn_stats)."""
    if training:
        return _bn_train_stats(c2, rows, C, bn[4], bn[5], bn[2], bn[3], bn[6] if len(bn) > 6 else None, parts=parts)
    mean, invstd = ops.bn_eval_params(bn[2], bn[3], bn[4])
    return mean, invstd, None


# A/B switch: 0 keeps the stand-alone statistics pass over every convolution output
_FUSE_BN_STATS = os.environ.get("AVSR_FUSE_BN_STATS", "1") != "0"


def _conv_bn_stats(x, wp, Cin, Cout, KH, KW, rows, training):
    """Mixed mode, split-plane components: a partial-statistics buffer for ops.conv2d_fwd(stats=...) -- the convolution's epilogue
    leaves per-column sums / sums of squares of every 128-row output tile, and the BatchNorm that follows is finished from those
    (one 44 us pass over a 200 MB f32 activation less per BatchNorm).  None: the convolution runs on a kernel without that
    epilogue, or the mode asks for the shifted two-level statistics of the stand-alone pass (precise / hpf parity at 1e-5)."""
    if not (training and _FUSE_BN_STATS and _state["mixed"] and ops.conv2d_takes_stats(x, wp, Cin, KH, KW, _state["precise"])):
        return None
    return torch.empty(ops.bn_stat_tiles(rows), 2, Cout, dtype=torch.float32, device=x.device)


def _bn_bwd(c, dy, add, mean, invstd, bn, counts, rows, C, act, want_dadd, training):
    """Backward of y = act(bn(c) + add): returns (dc, dadd, dgamma, dbeta)."""
    sums = ops.bn_bwd_reduce(c, dy, add, mean, invstd, bn[0], bn[1], rows, C, act)
    dgamma, dbeta = sums[1], sums[0]  # views of a fresh tensor
    if training:
        sums_dx, inv_n, n_dev = _bn_bwd_sums(sums, counts, rows)
    else:
        sums_dx, inv_n, n_dev = torch.zeros_like(sums), 0.0, None
    dc, dadd = ops.bn_bwd_apply(c, dy, add, mean, invstd, bn[0], bn[1], sums_dx, inv_n, rows, C, act, want_dadd,
                                n_dev=n_dev)
    return dc, dadd, dgamma, dbeta


def bn_tuple(m):
    """Pack a torch BatchNorm module for the front-end functions.  In training the batch counter is incremented by
    the statistics kernel (bn_finalize) -- one launch less per BatchNorm than `num_batches_tracked.add_(1)`."""
    return (m.weight, m.bias, m.running_mean, m.running_var, float(m.eps), float(m.momentum if m.momentum is not None else 0.1),
            m.num_batches_tracked if m.training else None)


_PRESPLIT = os.environ.get("AVSR_PRESPLIT", "1") != "0"  # A/B switch: split8 activations between the split-plane trunk stages


def _presplit_ok():
    """Is the running component one whose activations may leave in the split8 layout?  Mixed mode, split-plane arithmetic, forward
    pass, producer-side twins on (the backward pass reads the bf16 twin, never the split8 tensor)."""
    return bool(_PRESPLIT and _state["mixed"] and _state["precise"] and ops.SPLIT_FAST and ops.TWIN is not None
                and not _state.get("in_bwd", False))


class BasicBlockFn(torch.autograd.Function):
    """frontend/resnet.py:82-98 (and resnet1d.py:83-99 with H = 1) on a channels-last activation:
    conv3x3(stride) -> BN -> SiLU -> conv3x3 -> BN -> (+ identity | + BN(conv1x1(stride))) -> SiLU,
    forward and backward, every convolution an implicit MFMA GEMM, BatchNorm in batch-statistics mode."""

    @staticmethod
    def forward(ctx, x, dims, stride, training, w1, g1, b1, w2, g2, b2, wd, gd, bd, bn1, bn2, bnd):
        N, H, W, Cin = dims
        Cout = w1.shape[0]
        KH, KW = w1.shape[2], w1.shape[3]
        ph, pw = (KH - 1) // 2, (KW - 1) // 2
        T = act_dtype()
        pr = _state["precise"]
        wpl = 2 if _state["wp2"] else 1  # mixed mode, "f16x2" component: both planes of the f16 filter copies
        x_arg = x
        x = _act_in(x)  # hpf / mixed: the previous trunk function handed over its bf16 twin; compute on the f32 / f16 original
        OH, OW = ops.conv_out(H, KH, stride, ph), ops.conv_out(W, KW, stride, pw)
        rows = N * OH * OW
        bn1 = (g1, b1) + bn1
        bn2 = (g2, b2) + bn2
        wp1 = _w_conv_fwd(w1, x)
        wpd = _w_conv_fwd(wd, x) if wd is not None else None
        # Mixed mode, split-plane component (round 5): activations that only a split-plane convolution and element-wise passes read
        # travel in the split8 layout -- written that way by the BatchNorm + activation pass, staged by the convolution without its
        # in-LDS conversion pass (csrc/gemm_split.hip, ACV = 2: stage-1 convolution 392 -> 294 us, tools/microbench_presplit.py).
        ps = _presplit_ok()
        if isinstance(x, ops.Split8) and not (pr and isinstance(wp1, ops.Split8) and (wd is None or isinstance(wpd, ops.Split8))):
            x = ops.scale_dropout(x, torch.float32)  # (a consumer that cannot read the layout: one pass back to plain f32)
        st1 = _conv_bn_stats(x, wp1, Cin, Cout, KH, KW, rows, training)
        c1 = ops.conv2d_fwd(x, wp1, N, H, W, Cin, Cout, KH, KW, stride, ph, pw, pr, stats=st1, wp_planes=wpl)
        cd = md = idd = nd = None
        pair = wd is not None and training and _state["bn_sync"] is not None
        if pair:
            # cross-rank BatchNorm, down-sampling block: both convolutions read x, their two BatchNorms' statistics cross the
            # ranks in ONE all-gather (functional_convmod._bn_train_stats_pair)
            bnd = (gd, bd) + bnd
            std = _conv_bn_stats(x, wpd, Cin, Cout, 1, 1, rows, training)
            cd = ops.conv2d_fwd(x, wpd, N, H, W, Cin, Cout, 1, 1, stride, 0, 0, pr, stats=std, wp_planes=wpl)
            (m1, i1, n1), (md, idd, nd) = _bn_train_stats_pair(
                c1, cd, rows, Cout, (bn1[4], bn1[5], bn1[2], bn1[3], bn1[6] if len(bn1) > 6 else None),
                (bnd[4], bnd[5], bnd[2], bnd[3], bnd[6] if len(bnd) > 6 else None), parts_a=st1, parts_b=std)
        else:
            m1, i1, n1 = _bn_fwd_params(c1, rows, Cout, bn1, training, parts=st1)
        wp2 = _w_conv_fwd(w2, c1)
        a1 = ops.bn_act_fwd(c1, None, m1, i1, g1, b1, rows, Cout, 1, out_split8=ps and isinstance(wp2, ops.Split8))
        st2 = _conv_bn_stats(a1, wp2, Cout, Cout, KH, KW, rows, training)
        c2 = ops.conv2d_fwd(a1, wp2, N, OH, OW, Cout, Cout, KH, KW, 1, ph, pw, pr, stats=st2, wp_planes=wpl)
        m2, i2, n2 = _bn_fwd_params(c2, rows, Cout, bn2, training, parts=st2)
        if wd is not None:
            if not pair:
                bnd = (gd, bd) + bnd
                std = _conv_bn_stats(x, wpd, Cin, Cout, 1, 1, rows, training)
                cd = ops.conv2d_fwd(x, wpd, N, H, W, Cin, Cout, 1, 1, stride, 0, 0, pr, stats=std, wp_planes=wpl)
                md, idd, nd = _bn_fwd_params(cd, rows, Cout, bnd, training, parts=std)
            r = ops.bn_act_fwd(cd, None, md, idd, gd, bd, rows, Cout, 0)
        else:
            r = x
        out = ops.bn_act_fwd(c2, r, m2, i2, g2, b2, rows, Cout, 1, out_split8=ps and Cout % 64 == 0)
        sx = x_arg if (_state["hpf"] and x_arg.dtype == torch.bfloat16 and x_arg is not x) else _A(x)  # (the handed-over twin itself)
        ctx.save_for_backward(sx, _A(c1), _A(a1), _A(c2), _A(cd), _A(r) if wd is not None else None, w1, w2, wd, g1, b1, g2, b2,
                              gd, bd, m1, i1, n1, m2, i2, n2, md, idd, nd)
        ctx.meta = (dims, stride, training, (OH, OW), bn1[2:], bn2[2:], bnd[2:] if wd is not None else None)
        return _hand_over(out)

    @staticmethod
    @_bwd_mode
    def backward(ctx, dout):
        (x, c1, a1, c2, cd, r, w1, w2, wd, g1, b1, g2, b2, gd, bd, m1, i1, n1, m2, i2, n2, md, idd, nd) = ctx.saved_tensors
        dims, stride, training, (OH, OW), r1, r2, rd = ctx.meta
        N, H, W, Cin = dims
        Cout = w1.shape[0]
        KH, KW = w1.shape[2], w1.shape[3]
        ph, pw = (KH - 1) // 2, (KW - 1) // 2
        T = act_dtype()
        pr = _state["precise"]
        rows = N * OH * OW
        dout = _to_act(dout)
        if wd is None:
            r = x
        dc2, dr, dg2, db2 = _bn_bwd(c2, dout, r, m2, i2, (g2, b2) + r2, n2, rows, Cout, 1, True, training)
        dw2 = ops.conv2d_wgrad(dc2, a1, N, OH, OW, Cout, Cout, KH, KW, 1, ph, pw, pr, torch_layout=True)
        da1 = ops.conv2d_dgrad(dc2, _w_conv(w2, True), None, N, OH, OW, Cout, Cout, KH, KW, 1,
                               ph, pw, pr)
        dc1, _, dg1, db1 = _bn_bwd(c1, da1, None, m1, i1, (g1, b1) + r1, n1, rows, Cout, 1, False, training)
        dw1 = ops.conv2d_wgrad(dc1, x, N, H, W, Cin, Cout, KH, KW, stride, ph, pw, pr, torch_layout=True)
        dwd = dgd = dbd = None
        if wd is not None:
            dcd, _, dgd, dbd = _bn_bwd(cd, dr, None, md, idd, (gd, bd) + rd, nd, rows, Cout, 0, False, training)
            dwd = ops.conv2d_wgrad(dcd, x, N, H, W, Cin, Cout, 1, 1, stride, 0, 0, pr, torch_layout=True)
            skip = ops.conv2d_dgrad(dcd, _w_conv(wd, True), None, N, H, W, Cin, Cout, 1, 1,
                                    stride, 0, 0, pr)
        else:
            skip = dr
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad(dc1, _w_conv(w1, True), skip, N, H, W, Cin, Cout, KH, KW,
                                  stride, ph, pw, pr)
        return (dx, None, None, None, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd, None, None, None)


def basic_block(x, dims, stride, training, conv1, bn1, conv2, bn2, down):
    """x: channels-last [N,H,W,Cin] activation-dtype tensor; modules supply the parameters."""
    _state["tag_ok"] = torch.is_grad_enabled()
    t1, t2 = bn_tuple(bn1), bn_tuple(bn2)
    if down is not None:
        td = bn_tuple(down[1])
        return BasicBlockFn.apply(x, dims, stride, training, conv1.weight, t1[0], t1[1], conv2.weight, t2[0], t2[1],
                                  down[0].weight, td[0], td[1], t1[2:], t2[2:], td[2:])
    return BasicBlockFn.apply(x, dims, stride, training, conv1.weight, t1[0], t1[1], conv2.weight, t2[0], t2[1], None, None,
                              None, t1[2:], t2[2:], None)


class StemFn(torch.autograd.Function):
    """Single-input-channel stem: conv (temporal x spatial taps) -> BN -> SiLU -> optional 3x3/s2 max-pool.
    Video: frontend/resnet.py:203-219 (Conv3d(1,64,(5,7,7),s(1,2,2)) + BatchNorm3d + SiLU + MaxPool3d).
    Audio: frontend/resnet1d.py:124-139,190-192 (Conv1d(1,64,80,s4) + BatchNorm1d + SiLU; no pooling)."""

    @staticmethod
    def forward(ctx, x, w, g, b, bn_rest, geom, pool, training):
        B, Tn, H, W, KT, KH, KW, stride, pt, ph, pw = geom
        Cout = w.shape[0]
        T = act_dtype()
        pr = _state["precise"]
        x = x.contiguous()
        st0 = None
        taps = KT * KH * KW
        ldw = padded_cols(taps)
        geom_ok = (KT, KH, KW, stride, pt, ph, pw, Cout) == (5, 7, 7, 2, 2, 3, 3, 64) and W % 4 == 0 and W <= 96 \
            and (W - 1) // 2 + 1 <= 64
        dedicated = geom_ok and not pr
        if dedicated:  # csrc/stem.hip: input rows staged once in LDS
            c0 = ops.stem357_fwd(x, w, B, Tn, H, W)
        elif geom_ok and pr and ops.SPLIT_FAST and x.dtype == torch.float32 and w.dtype == torch.float32:
            # the same kernel on split hi / lo planes, f32 result; mixed mode: it leaves the BatchNorm statistics of its output
            # behind (the stand-alone pass over this 790 MB activation took 175 us)
            if training and _FUSE_BN_STATS and _state["mixed"]:
                c0, st0 = ops.stem357_fwd_f32s(x, w.contiguous(), B, Tn, H, W, want_stats=True)
            else:
                c0 = ops.stem357_fwd_f32s(x, w.contiguous(), B, Tn, H, W)
        else:
            wp = ops.conv_weight_permute(w, T, ld_out=ldw)
            c0 = ops.conv_stem_fwd(x, wp, ldw, T, B, Tn, H, W, Cout, KT, KH, KW, stride, pt, ph, pw, pr)
        OH, OW = c0.shape[1], c0.shape[2]
        rows = B * Tn * OH * OW
        bn = (g, b) + bn_rest
        m0, i0, n0 = _bn_fwd_params(c0, rows, Cout, bn, training, parts=st0)
        idx = xsel = None
        if pool and AF._FUSE_STEM_POOL:
            # BN + SiLU + max-pool in one pass: the full-resolution activation (396 MB per 1600 video frames) is never
            # written (the backward pass recomputes it from c0 anyway)
            # xsel: the raw conv output at every arg-max -- all the backward reduce pass needs of c0
            out, idx, xsel = ops.bn_act_pool_fwd(c0, m0, i0, g, b, B * Tn, OH, OW, Cout, 3, 2, 1, 1, want_xsel=True,
                                                 out_split8=_presplit_ok() and c0.dtype == torch.float32 and Cout % 64 == 0
                                                 and AF.MIXED_POLICY.get("trunk1", "split") == "split")
        elif pool:
            a0 = ops.bn_act_fwd(c0, None, m0, i0, g, b, rows, Cout, 1)
            out, idx = ops.maxpool2d_fwd(a0, B * Tn, OH, OW, Cout, 3, 2, 1)
        else:
            out = ops.bn_act_fwd(c0, None, m0, i0, g, b, rows, Cout, 1)
        ctx.save_for_backward(x, _A(c0), idx, g, b, m0, i0, n0, _A(xsel))
        ctx.meta = (geom, pool, training, bn_rest, (OH, OW), w.shape, geom_ok and not _bwd_precise())
        if _state["hpf"] and out.dtype == torch.float32 and out.data_ptr() not in _twins and _state.get("tag_ok", True):
            # (the pooled output has no producer-side twin: make it here -- the first residual block would cast it anyway)
            _twins[out.data_ptr()] = (out, ops.scale_dropout(out, torch.bfloat16))
        return _hand_over(out)

    @staticmethod
    @_bwd_mode
    def backward(ctx, dout):
        x, c0, idx, g, b, m0, i0, n0, xsel = ctx.saved_tensors
        geom, pool, training, bn_rest, (OH, OW), wshape, dedicated = ctx.meta
        B, Tn, H, W, KT, KH, KW, stride, pt, ph, pw = geom
        Cout = wshape[0]
        rows = B * Tn * OH * OW
        dout = _to_act(dout)
        if pool and AF._FUSE_STEM_POOL:
            # the activation gradient is gathered from the pooled gradient inside both BatchNorm backward passes: the
            # full-resolution gradient (396 MB per 1600 video frames) is neither written nor read back
            dp = _to_act(dout)
            # sum over pixels of dz == sum over pooled outputs of dpool * act'(z(arg-max pixel)): the reduce pass runs on
            # the pooled tensors (a quarter of the pixels) and never reads c0
            POH, POW = ops.conv_out(OH, 3, 2, 1), ops.conv_out(OW, 3, 2, 1)
            sums = ops.bn_bwd_reduce(xsel, dp, None, m0, i0, g, b, B * Tn * POH * POW, Cout, 1)
            dg, db = sums[1], sums[0]
            if training:
                sums_dx, inv_n, n_dev = _bn_bwd_sums(sums, n0, rows)
            else:
                sums_dx, inv_n, n_dev = torch.zeros_like(sums), 0.0, None
            dc0 = ops.bn_pool_bwd_apply(c0, dp, idx, m0, i0, g, b, sums_dx, inv_n, B * Tn, OH, OW, Cout, 3, 2, 1, 1,
                                        n_dev=n_dev)
        else:
            da0 = ops.maxpool2d_bwd(idx, dout, B * Tn, OH, OW, Cout, 3, 2, 1) if pool else dout
            dc0, _, dg, db = _bn_bwd(c0, da0, None, m0, i0, (g, b) + bn_rest, n0, rows, Cout, 1, False, training)
        if dedicated:
            dw = ops.stem357_wgrad(dc0, x, B, Tn, H, W)
        else:
            dw = ops.conv_stem_wgrad(dc0, x, B, Tn, H, W, Cout, KT, KH, KW, stride, pt, ph, pw, _state["precise"])
        return None, dw.view(wshape), dg, db, None, None, None, None


def stem(x, conv, bn, geom, pool):
    _state["tag_ok"] = torch.is_grad_enabled()
    t = bn_tuple(bn)
    return StemFn.apply(x, conv.weight, t[0], t[1], t[2:], geom, pool, bn.training)


class AvgPoolFn(torch.autograd.Function):
    """Mean over groups of `win` consecutive pixels of a channels-last tensor -> f32 [groups, C]
    (AdaptiveAvgPool2d(1), resnet.py:117,164; AvgPool1d(20), resnet1d.py:143-146)."""

    @staticmethod
    def forward(ctx, x, groups, win, C):
        ctx.meta = (groups, win, C, x.dtype, x.shape)
        xi = _f32_in(x)  # (hpf: the trunk hands over its bf16 twin)
        if isinstance(xi, ops.Split8):
            xi = ops.scale_dropout(xi, torch.float32)
        return ops.avgpool_fwd(xi.contiguous(), groups, win, C)

    @staticmethod
    @_bwd_mode
    def backward(ctx, dy):
        groups, win, C, dtype, shape = ctx.meta
        return ops.avgpool_bwd(_to_f32(dy), dtype, groups, win, C).view(shape), None, None, None


def avg_pool(x, groups, win, C):
    _state["tag_ok"] = torch.is_grad_enabled()
    return AvgPoolFn.apply(x, groups, win, C)
