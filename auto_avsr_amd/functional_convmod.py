"""Autograd glue of the Conformer convolution module (split out of functional.py in round 4): BatchNorm statistics plumbing
(single-rank / cross-rank) and ConvSublayerFn.  Re-exported by functional.py."""

import torch

from . import functional as AF
from . import ops
from .functional import (  # noqa: F401
    _A, _bwd_mode, _chain_tag, _chain_take, _drop_args, _gemm_nn, _gemm_nt, _ln_bwd, _prologue, _state, _to_act,
    _to_f32, _wgrad, _zeros, act_dtype)


# ------------------------------------------------------------------------------------------------ BatchNorm plumbing
_const_cache = {}


def _const1(value, device):
    """A resident 1-element f32 constant (BatchNorm row counts): one fill per distinct value instead of one per
    BatchNorm per step.  Never created under hipGraph capture (its fill would only run at replay)."""
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        t = _const_cache.get((device, value))
        return t if t is not None else torch.full((1,), value, dtype=torch.float32, device=device)
    key = (device, value)
    t = _const_cache.get(key)
    if t is None:
        if len(_const_cache) > 4096:
            _const_cache.clear()
        t = _const_cache[key] = torch.full((1,), value, dtype=torch.float32, device=device)
    return t


def _bn_train_stats(c2, rows, C, eps, momentum, running_mean, running_var, nbt=None, parts=None):
    """Batch statistics (+ running-stat update, + num_batches_tracked count) of a [rows, C] activation; merged across
    ranks when set_bn_sync().  parts: the partial statistics the producing convolution's epilogue left behind
    (ops.conv2d_fwd(stats=...)) -- no pass over c2 then."""
    group = _state["bn_sync"]
    if group is not None:
        import torch.distributed as dist

        # one all-gather of {shifted statistics, row count} per BatchNorm (ranks hold different row counts); the
        # payload is written by the statistics kernel and read in place (strided) by the merge kernel, which also
        # leaves the global row count on the device for the backward pass -- no glue launches around the collective
        comm = _state.get("bn_comm")
        W = comm.world if comm is not None else dist.get_world_size(group)
        mine = ops.bn_stats_parts(parts, rows, C) if parts is not None else ops.bn_stats(c2, rows, C, with_count=True)
        flat = torch.empty(W * mine.numel(), dtype=torch.float32, device=c2.device)
        if comm is not None:
            comm.all_gather(flat, mine)
        else:
            dist.all_gather_into_tensor(flat, mine, group=group)
        n_total = torch.empty(1, dtype=torch.float32, device=c2.device)
        mean, invstd = ops.bn_finalize(flat, flat.data_ptr() + 12 * C, W, C, eps, momentum, running_mean, running_var,
                                       nbt, stats_stride=3 * C + 1, counts_stride=3 * C + 1, n_total=n_total)
        return mean, invstd, n_total
    if parts is not None:
        mean, invstd = ops.bn_finalize_parts(parts, rows, C, eps, momentum, running_mean, running_var, nbt)
    else:
        mean, invstd = ops.bn_stats_finalize(c2, rows, C, eps, momentum, running_mean, running_var, nbt)
    return mean, invstd, None


def _bn_train_stats_pair(ca, cb, rows, C, bn_a, bn_b, parts_a=None, parts_b=None):
    """Cross-rank batch statistics of TWO independent BatchNorms over activations of the same shape -- the main path's first
    BatchNorm and the down-sampling path's of a residual block (resnet.py:82-98, resnet1d.py:83-99: both convolutions read the
    block input) -- through ONE all-gather: the two {statistics, row count} payloads travel side by side and each merge kernel reads
    its half of the gathered buffer strided.  One latency-bound collective less per down-sampling block and step (train.py:31
    `sync_batchnorm=True` costs 64 of them).  bn_x = (eps, momentum, running_mean, running_var, num_batches_tracked)."""
    import torch.distributed as dist

    group = _state["bn_sync"]
    assert group is not None
    comm = _state.get("bn_comm")
    W = comm.world if comm is not None else dist.get_world_size(group)
    mine = torch.cat([ops.bn_stats_parts(p, rows, C) if p is not None else ops.bn_stats(c, rows, C, with_count=True)
                      for c, p in ((ca, parts_a), (cb, parts_b))])
    P = 3 * C + 1
    flat = torch.empty(W * 2 * P, dtype=torch.float32, device=ca.device)
    if comm is not None:
        comm.all_gather(flat, mine)
    else:
        dist.all_gather_into_tensor(flat, mine, group=group)
    out = []
    for k, (eps, momentum, rm, rv, nbt) in enumerate((bn_a, bn_b)):
        half = flat[k * P:]
        n_total = torch.empty(1, dtype=torch.float32, device=ca.device)
        mean, invstd = ops.bn_finalize(half, half.data_ptr() + 12 * C, W, C, eps, momentum, rm, rv, nbt,
                                       stats_stride=2 * P, counts_stride=2 * P, n_total=n_total)
        out.append((mean, invstd, n_total))
    return out


def _bn_bwd_sums(sums, counts, rows):
    """All-reduce the backward sums across the sync group; returns (sums_for_dx, inv_n, n_dev).  `counts` is what
    _bn_train_stats returned: under synchronisation the global row count, resident on the device."""
    group = _state["bn_sync"]
    if group is None:
        return sums, 1.0 / rows, None
    import torch.distributed as dist

    tot = sums.clone()
    if _state.get("bn_comm") is not None:
        _state["bn_comm"].all_reduce(tot)
    else:
        dist.all_reduce(tot, group=group)
    return tot, 0.0, counts  # global row count stays on the device (no host sync)


class ConvSublayerFn(torch.autograd.Function):
    """x + dropout(ConvolutionModule(LN(x))):  conformer_encoder.py:145-151,30-35.
    pointwise(D->2D) -> GLU -> depthwise(K) -> BatchNorm1d (batch stats over every frame) -> SiLU -> pointwise."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w_pw1, b_pw1, w_dw, b_dw, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, w_pw2, b_pw2, training,
                momentum, bn_eps, p_out, eps):
        x = x.contiguous()
        B, Tn, D = x.shape
        rows = B * Tn
        K = w_dw.shape[-1]
        T = act_dtype()
        fused = ln_w is not None  # False: bare ConvolutionModule.forward (no LayerNorm, no residual)
        ctx.chain = _chain_take(x) if fused else None
        if fused:
            h, mean, rstd = ops.layernorm_fwd(x, ln_w, ln_b, T, eps, twin=True)
        else:
            h, mean, rstd = _to_act(x), None, None
        # Mixed mode, f16 component (round 5): what only ELEMENT-WISE kernels read -- the pointwise-1 output (GLU + depthwise
        # convolution) and the depthwise output (BatchNorm + Swish) -- stays f32; the chain is rounded to f16 once, by the
        # BatchNorm kernel, where the next consumer (pointwise 2) is an MFMA operand.  Two f16 roundings per layer fewer on the
        # residual branch the parity study found most sensitive (tools/precision_study.py "enc_conv"); the backward pass reads
        # the bf16 twins the producers write either way.
        Tc = torch.float32 if _state["f16"] else T
        a = torch.empty(rows, 2 * D, dtype=Tc, device=x.device)
        _gemm_nt(h, w_pw1.view(2 * D, D), rows, 2 * D, D, a, bias=b_pw1, twin=True)
        # GLU (conformer_encoder.py:32) is folded into the depthwise convolution: its window staging forms
        # a[:, :D] * sigmoid(a[:, D:]) on the fly, the GLU output is never written
        gl = None
        wdw = w_dw.view(D, K)
        one_launch = training and AF._BN_SMALL and _state["bn_sync"] is None and rows <= ops.BN_SMALL_MAX_ROWS
        # round 6 experiment (AVSR_CONVMOD_FUSED=1, off by default: slower, functional.py _CONVMOD_FUSED): the whole element-wise
        # middle (GLU, depthwise conv, BatchNorm statistics + running stats + normalise, Swish) as ONE launch -- a block owns 8
        # channels and every frame (csrc/convmod_fused.hip)
        middle = one_launch and AF._CONVMOD_FUSED
        ctx.middle = middle
        c = None if middle else ops.dwconv(a, wdw, b_dw, B, Tn, D, K, glu_in=True)
        if middle:
            s, c, bmean, binv = ops.convmod_dwbn_fwd(a, wdw, b_dw, B, Tn, D, K, bn_w, bn_b, bn_eps, momentum, bn_rm, bn_rv, bn_nbt,
                                                     out_dtype=T if Tc != T else None)
            counts = None
        elif one_launch:  # statistics + running stats + normalise + Swish in one pass
            s, bmean, binv = ops.bn_small_fwd(c, rows, D, bn_w, bn_b, bn_eps, momentum, bn_rm, bn_rv, bn_nbt, 1,
                                              out_dtype=T if Tc != T else None)
            counts = None
        else:
            if training:
                bmean, binv, counts = _bn_train_stats(c, rows, D, bn_eps, momentum, bn_rm, bn_rv, bn_nbt)
            else:
                bmean, binv = ops.bn_eval_params(bn_rm, bn_rv, bn_eps)
                counts = None
            s = ops.bn_act_fwd(c, None, bmean, binv, bn_w, bn_b, rows, D, 1)
            if s.dtype != T:  # (f32 chain of the mixed mode on the cross-rank / evaluation path: one cast launch)
                s = _to_act(s)
        po, so, sdo = _drop_args(p_out, x)
        y = torch.empty_like(x)
        _gemm_nt(s, w_pw2.view(D, D), rows, D, D, y, bias=b_pw2, drop_p=po, seed=so, seed_dev=sdo,
                 resid=x if fused else None, ldr=D)
        ctx.save_for_backward(x, ln_w, mean, rstd, _A(h), _A(a), gl, _A(c), bmean, binv, bn_w, bn_b, _A(s), w_pw1, wdw, w_pw2, counts)
        ctx.meta = (training, po, so, sdo, K, fused)
        if fused:
            _chain_tag(y, rows, D, 1.0, (po, so, sdo))
        return y

    @staticmethod
    @_bwd_mode
    def backward(ctx, dy):
        (x, ln_w, mean, rstd, h, a, gl, c, bmean, binv, bn_w, bn_b, s, w_pw1, wdw, w_pw2, counts) = ctx.saved_tensors
        training, po, so, sdo, K, fused = ctx.meta
        dy = dy.contiguous()
        B, Tn, D = x.shape
        rows = B * Tn
        T = act_dtype()
        g, gT, _ = _prologue(dy, rows, D, drop=(po, so, sdo), want_bias=False)
        db2 = _zeros(D, x.device)
        ds = torch.empty(rows, D, dtype=T, device=x.device)
        with ops.paired():
            dw2 = _wgrad(g, s, rows, D, D, bias_out=db2).view(D, D, 1)
            _gemm_nn(g, w_pw2.view(D, D), rows, D, D, ds)
        dwdw = _zeros((D, K), x.device)
        dbdw = _zeros(D, x.device)
        if ctx.middle:  # BatchNorm + Swish backward, depthwise weight / data gradient, GLU backward: one launch
            da, dbn_w, dbn_b = ops.convmod_dwbn_bwd(a, c, ds, bmean, binv, bn_w, bn_b, wdw, B, Tn, D, K, dwdw, dbdw)
            dc = None
        elif training and AF._BN_SMALL and _state["bn_sync"] is None and rows <= ops.BN_SMALL_MAX_ROWS:
            dc, dbn_w, dbn_b = ops.bn_small_bwd(c, ds, rows, D, bmean, binv, bn_w, bn_b, 1)
        else:
            sums = ops.bn_bwd_reduce(c, ds, None, bmean, binv, bn_w, bn_b, rows, D, 1)
            dbn_w, dbn_b = sums[1], sums[0]
            if training:
                sums_dx, inv_n, n_dev = _bn_bwd_sums(sums, counts, rows)
            else:
                sums_dx, inv_n, n_dev = torch.zeros_like(sums), 0.0, None
            dc, _ = ops.bn_bwd_apply(c, ds, None, bmean, binv, bn_w, bn_b, sums_dx, inv_n, rows, D, 1, False, n_dev=n_dev)
        if dc is not None:
            ops.dwconv_wgrad(a, dc, dwdw, dbdw, B, Tn, D, K, glu_in=True)
            # data gradient of the depthwise convolution with the GLU backward as its epilogue: d glu never reaches HBM
            da = ops.dwconv(dc, wdw, None, B, Tn, D, K, flip=True, glu_a=a).view(rows, 2 * D)
        db1 = _zeros(2 * D, x.device)
        if fused:
            dh = torch.empty(rows, D, dtype=T, device=x.device)
            with ops.paired():
                dw1 = _wgrad(da, h, rows, 2 * D, D, bias_out=db1).view(2 * D, D, 1)
                _gemm_nn(da, w_pw1.view(2 * D, D), rows, D, 2 * D, dh)
            dg = _zeros(D, x.device)
            dbt = _zeros(D, x.device)
            dx = _ln_bwd(dh, x, ln_w, mean, rstd, dg, dbt, dy, ctx.chain)
        else:
            dg = dbt = None
            dx = torch.empty(B, Tn, D, dtype=torch.float32, device=x.device)
            with ops.paired():
                dw1 = _wgrad(da, h, rows, 2 * D, D, bias_out=db1).view(2 * D, D, 1)
                _gemm_nn(da, w_pw1.view(2 * D, D), rows, D, 2 * D, dx)
        return (dx, dg, dbt, dw1, db1, dwdw.view(D, 1, K), dbdw, dbn_w, dbn_b, None, None, None, dw2, db2, None, None,
                None, None, None)


def conv_sublayer(x, ln_w, ln_b, w_pw1, b_pw1, w_dw, b_dw, bn, w_pw2, b_pw2, p_out, eps=1e-12):
    """bn: the torch.nn.BatchNorm1d module holding weight / bias / running stats (updated in place in training).
    ln_w = ln_b = None gives the bare module (no LayerNorm, no residual, no output dropout)."""
    training = bn.training
    _state["tag_ok"] = torch.is_grad_enabled()
    momentum = bn.momentum if bn.momentum is not None else 0.1
    # the batch counter of the BatchNorm is incremented by its statistics kernel (bn_finalize)
    return ConvSublayerFn.apply(_to_f32(x), ln_w, ln_b, w_pw1, b_pw1, w_dw, b_dw, bn.weight, bn.bias, bn.running_mean,
                                bn.running_var, bn.num_batches_tracked if training else None, w_pw2, b_pw2, training,
                                float(momentum), float(bn.eps), float(p_out), eps)
