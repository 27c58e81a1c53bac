"""Autograd glue of the small heads (split out of functional.py in round 4): scale + dropout, embedding, CTC loss,
label-smoothing cross entropy, row additions, log-softmax.  Re-exported by functional.py."""

import torch

from . import functional as AF
from . import ops
from .functional import (  # noqa: F401
    _bwd_mode, _drop_args, _pitched_2d, _to_f32, act_dtype, padded_cols)


# ------------------------------------------------------------------------------------------------ misc
class ScaleDropoutFn(torch.autograd.Function):
    """alpha * dropout(x) -> f32   (embedding.py:179-184: x*sqrt(d) then dropout; ctc.py:54 dropout)."""

    @staticmethod
    def forward(ctx, x, alpha, p, out_dtype):
        pp, s, sd = _drop_args(p, x)
        ctx.meta = (alpha, pp, s, sd, x.dtype)
        return ops.scale_dropout(x.contiguous(), out_dtype, alpha=alpha, drop_p=pp, seed=s, seed_dev=sd)

    @staticmethod
    @_bwd_mode
    def backward(ctx, dy):
        alpha, pp, s, sd, in_dtype = ctx.meta
        out = torch.float32 if in_dtype == torch.float32 else act_dtype()
        return ops.scale_dropout(dy.contiguous(), out, alpha=alpha, drop_p=pp, seed=s, seed_dev=sd), None, None, None


def scale_dropout(x, alpha=1.0, p=0.0, out_dtype=torch.float32):
    return ScaleDropoutFn.apply(x, float(alpha), float(p), out_dtype)


class EmbedFn(torch.autograd.Function):
    """dropout(table[ids]*sqrt(d) + pe[pos])   transformer_decoder.py:186-189 + embedding.py:78-87."""

    @staticmethod
    def forward(ctx, ids, table, pe, scale, p):
        L = ids.shape[-1]
        pp, s, sd = _drop_args(p, table)
        ids = ids.contiguous()
        ctx.save_for_backward(ids)
        ctx.meta = (scale, pp, s, sd, table.shape)
        return ops.embed_fwd(ids, table, pe[:L].contiguous(), L, scale, pp, s, sd)

    @staticmethod
    @_bwd_mode
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        scale, pp, s, sd, tshape = ctx.meta
        dt = torch.zeros(tshape, dtype=torch.float32, device=dy.device)
        ops.embed_bwd(ids, _to_f32(dy), dt, scale, pp, s, sd)
        return None, dt, None, None, None


def embed(ids, table, pe, scale, p):
    return EmbedFn.apply(ids, table, pe, float(scale), float(p))


# ------------------------------------------------------------------------------------------------ loss heads
class CtcLossFn(torch.autograd.Function):
    """ctc.py:32-38: log_softmax + CTCLoss(sum, zero_infinity) / B on f32 logits [B,T,V] (possibly a [..., :V]
    view of a pitch-padded buffer).  The gradient is produced by the forward kernels."""

    @staticmethod
    def forward(ctx, logits, labels, in_lens, ignore_id):
        B, Tn, V = logits.shape
        pit = _pitched_2d(logits, B * Tn, V)
        if pit is None or logits.dtype != torch.float32:
            ld = padded_cols(V)
            buf = torch.zeros(B * Tn, ld, dtype=torch.float32, device=logits.device)
            buf[:, :V].copy_(logits.reshape(B * Tn, V))
            pit = (buf, ld)
        l2, ld = pit
        lab = labels.reshape(B, -1).contiguous()
        nll, grad = ops.ctc_loss(l2, ld, lab, in_lens.to(torch.int64).contiguous(), B, Tn, V,
                                 want_grad=True, ignore_id=ignore_id)
        loss = ops.sum_finite_scale(nll, 1.0 / B)
        ctx.save_for_backward(grad)
        ctx.meta = (B, Tn, V, ld)
        return loss.view(())

    @staticmethod
    @_bwd_mode
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        B, Tn, V, ld = ctx.meta
        d = ops.scale_dropout(grad, torch.float32, alpha=1.0 / B, alpha_dev=g.reshape(1).to(torch.float32).contiguous())
        return d.view(B, Tn, ld)[..., :V], None, None, None


def ctc_loss(logits, labels, in_lens, ignore_id=-1):
    return CtcLossFn.apply(logits, labels, in_lens, ignore_id)


class CeSmoothFn(torch.autograd.Function):
    """label_smoothing_loss.py:41-63 (sum over tokens / B) + nets_utils.py:272-292 accuracy, on f32 logits
    [B,L,V].  Returns (loss, n_hits, n_valid) as device scalars."""

    @staticmethod
    def forward(ctx, logits, target, smoothing, ignore_id, denom):
        V = logits.shape[-1]
        rows = logits.numel() // V
        pit = _pitched_2d(logits, rows, V)
        if pit is None or logits.dtype != torch.float32:
            ld = padded_cols(V)
            buf = torch.zeros(rows, ld, dtype=torch.float32, device=logits.device)
            buf[:, :V].copy_(logits.reshape(rows, V))
            pit = (buf, ld)
        l2, ld = pit
        tgt = target.reshape(-1).to(torch.int64).contiguous()
        row_loss, row_hit, grad = ops.ce_smooth(l2, ld, tgt, V, smoothing, want_grad=True, ignore_id=ignore_id)
        loss = ops.sum_scale(row_loss, 1.0 / denom)
        hits = ops.sum_scale(row_hit, 1.0)
        ctx.save_for_backward(grad)
        ctx.meta = (logits.shape, ld, denom)
        ctx.mark_non_differentiable(hits)
        return loss.view(()), hits.view(())

    @staticmethod
    @_bwd_mode
    def backward(ctx, g, _gh):
        (grad,) = ctx.saved_tensors
        shape, ld, denom = ctx.meta
        d = ops.scale_dropout(grad, torch.float32, alpha=1.0 / denom,
                              alpha_dev=g.reshape(1).to(torch.float32).contiguous())
        return d.view(shape[:-1] + (ld,))[..., : shape[-1]], None, None, None, None


def ce_smooth(logits, target, smoothing, ignore_id, denom):
    return CeSmoothFn.apply(logits, target, float(smoothing), int(ignore_id), float(denom))


class AddRowsFn(torch.autograd.Function):
    """dropout(x*scale + table[t])  for x (B, n, D), table (n, D) -- embedding.py:78-87 as a stand-alone module."""

    @staticmethod
    def forward(ctx, x, table, scale, p):
        pp, s, sd = _drop_args(p, x)
        ctx.meta = (scale, pp, s, sd)
        return ops.scale_dropout(x.contiguous(), torch.float32, alpha=scale, drop_p=pp, seed=s, seed_dev=sd,
                                 add=table, add_period=table.numel())

    @staticmethod
    @_bwd_mode
    def backward(ctx, dy):
        scale, pp, s, sd = ctx.meta
        return ops.scale_dropout(dy.contiguous(), torch.float32, alpha=scale, drop_p=pp, seed=s, seed_dev=sd), None, None, None


def add(a, b):
    """a + b (f32 result; inference-time glue of the incremental decoder, no autograd)."""
    bb = _to_f32(b)
    return ops.scale_dropout(a.contiguous(), torch.float32, add=bb, add_period=bb.numel())


def log_softmax(logits):
    """Row-wise log-softmax of f32 logits [..., V] (possibly a [..., :V] view of a pitch-padded buffer); inference
    helper of ctc.py:76-83 and transformer_decoder.py:288."""
    V = logits.shape[-1]
    rows = logits.numel() // V
    pit = _pitched_2d(logits, rows, V)
    if pit is None or logits.dtype != torch.float32:
        ld = padded_cols(V)
        buf = torch.zeros(rows, ld, dtype=torch.float32, device=logits.device)
        buf[:, :V].copy_(logits.reshape(rows, V))
        pit = (buf, ld)
    l2, ld = pit
    out = ops.log_softmax(l2, ld, rows, V)
    return out.view(logits.shape[:-1] + (ld,))[..., :V]
