"""Autograd glue of the attention sub-layers (split out of functional.py in round 4): the all-layer position projection
(PosProjFn / prepare_pos_proj), AttentionCoreFn, MhaSublayerFn (rel-pos self attention, decoder self / source attention) and
the all-layer K / V projection of the encoder memory (MemoryKVFn).  State, numerical modes, weight caches and the shared
helpers live in functional.py, which re-exports everything defined here."""
import math

import torch

from . import functional as AF
from . import ops
from .functional import (  # noqa: F401
    _A, _A_shared, _A_view, _bgrad, _bias3, _bwd_mode, _chain_tag, _chain_take, _drop_args, _gemm_nn, _gemm_nt,
    _ln_bwd, _mask_arg, _pos_proj, _prologue, _state, _to_act, _to_act_shared, _to_f32, _w_bf16_cat, _w_h16_cat, _h16_nt,
    _wgrad, _zeros, act_dtype)


def prepare_pos_proj(pos_emb, weights):
    """bf16 mode: pos_emb [1, P, D] @ [W_0; W_1; ...]^T -> [P, n*D] in one launch; layer l's relative-position attention then
    reads its D-column block in place (row pitch n*D) instead of running its own P x D x D projection.  The weight gradients
    stay per layer (dW_l = dpos_l^T pe).  No-op in precise mode / for shapes the tuned kernel does not take -- the layers
    then project on their own.  Entries are dropped by new_step()."""
    _pos_proj.clear()
    if _state["precise"] or len(weights) < 2 or pos_emb is None:
        return
    D = pos_emb.shape[-1]
    if D % 64 or any(w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != (D, D) for w in weights):
        return
    pe = _to_act_shared(pos_emb).reshape(-1, D)
    if pe.dtype not in (torch.bfloat16, torch.float16):
        return
    n, P = len(weights), pe.shape[0]
    if torch.is_grad_enabled() and all(w.requires_grad for w in weights):
        # training: the projection is an autograd node of its own, so that the n weight gradients are ONE contraction too
        holder = {}
        out = PosProjFn.apply(pos_emb, holder, *weights)
        for i, w in enumerate(weights):
            _pos_proj[(pos_emb.data_ptr(), w.data_ptr())] = (pos_emb, None, out, i, holder)
        return
    out = torch.empty(P, n * D, dtype=pe.dtype, device=pe.device)
    if pe.dtype == torch.float16:
        _h16_nt(pe, D, _w_h16_cat(tuple(weights)), P, n * D, D, out, n * D, planes=1)  # (one plane: see PosProjFn)
    else:
        ops.gemm_bf16_nt(pe, D, _w_bf16_cat(tuple(weights), False), D, P, n * D, D, out, n * D)
    for i, w in enumerate(weights):
        _pos_proj[(pos_emb.data_ptr(), w.data_ptr())] = (pos_emb, out[:, i * D:(i + 1) * D], None, i, None)


_placeholders = {}


def _placeholder_grad(shape, dtype, device):
    """A zero 'gradient' of the right shape / dtype that costs no launch and no memory (an expanded scalar): tells autograd
    that the producer's backward may run, while the real gradient sits in a side buffer."""
    z = _placeholders.get((dtype, device))
    if z is None:
        z = _placeholders[(dtype, device)] = torch.zeros((), dtype=dtype, device=device)
    return z.expand(shape)


class PosProjFn(torch.autograd.Function):
    """linear_pos of every encoder layer applied to the (batch-shared, layer-independent) position table
    (attention.py:170): forward one [P, D] x [D, n*D] GEMM; backward one [n*D, D] = dpos_all^T pe contraction over the
    buffer whose column blocks the layers' attention backward accumulated into (MhaSublayerFn, ctx.pp) -- instead of n
    zero-fills and n 144-tile GEMMs that each leave half of the chip idle."""

    @staticmethod
    def forward(ctx, pos_emb, holder, *weights):
        D = pos_emb.shape[-1]
        pe = _to_act_shared(pos_emb).reshape(-1, D)
        n, P = len(weights), pe.shape[0]
        out = torch.empty(P, n * D, dtype=pe.dtype, device=pe.device)
        if pe.dtype == torch.float16:  # mixed mode: f16 projection + its bf16 twin (the layers' backward passes read views of it)
            # the hi plane of linear_pos alone: the position term sits behind the softmax, where operand rounding averages out
            # (tools/precision_study.py: q / k / pos projections on one plane leave the logits error unchanged)
            _h16_nt(pe, D, _w_h16_cat(tuple(weights)), P, n * D, D, out, n * D, planes=1, twin=True)
        else:
            ops.gemm_bf16_nt(pe, D, _w_bf16_cat(tuple(weights), False), D, P, n * D, D, out, n * D)
        ctx.save_for_backward(_A_shared(pe))
        ctx.holder, ctx.meta = holder, (P, D, n)
        return out

    @staticmethod
    @_bwd_mode
    def backward(ctx, dout):
        (pe,) = ctx.saved_tensors
        P, D, n = ctx.meta
        holder = ctx.holder
        assert holder.get("filled", 0) == n and holder.get("dpos") is not None, \
            "PosProjFn: every encoder layer must have accumulated its position gradient"
        # (`dout` is a placeholder: autograd would round a real f32 gradient to the bf16 of the forward output)
        dW = _wgrad(holder["dpos"], pe, P, n * D, D)
        holder["dpos"] = None
        holder["filled"] = 0
        return (None, None) + tuple(dW[i * D:(i + 1) * D] for i in range(n))



# ------------------------------------------------------------------------------------------------ attention cores
def _proj(h, w, b, rows, D, twin=True):
    out = torch.empty(rows, w.shape[0], dtype=act_dtype(), device=h.device)
    _gemm_nt(h, w, rows, w.shape[0], D, out, bias=b, twin=twin)
    return out


class AttentionCoreFn(torch.autograd.Function):
    """Projections + fused attention + output projection of attention.py:90-104 / 153-193, *without* residual:
    returns linear_out(softmax(...) v) as f32.  q_in / kv_in are activation-dtype or f32 (rows x D) inputs."""

    @staticmethod
    def forward(ctx, q_in, kv_in, pos_emb, mask, wq, bq, wk, bk, wv, bv, wo, bo, wpos, bias_u, bias_v, H, p_attn,
                same_kv):
        B, Tq, D = q_in.shape
        Tk = kv_in.shape[1]
        dk = D // H
        T = act_dtype()
        qa = _to_act(q_in)
        ka = qa if same_kv else _to_act(kv_in)
        q = _proj(qa, wq, bq, B * Tq, D, twin=pos_emb is None)
        k = _proj(ka, wk, bk, B * Tk, D)
        v = _proj(ka, wv, bv, B * Tk, D)
        relpos = pos_emb is not None
        pe = pproj = qv = None
        if relpos:
            pe = _to_act(pos_emb.reshape(-1, D))
            pproj = torch.empty(pe.shape[0], D, dtype=T, device=q.device)
            _gemm_nt(pe, wpos, pe.shape[0], D, D, pproj)
            qu, qv = ops.head_bias_fwd(q, D, B * Tq, D, bias_u.reshape(-1), bias_v.reshape(-1))
        else:
            qu = q
        pa, sa, sda = _drop_args(p_attn, q_in)
        m = _mask_arg(mask)
        ctxv, lse = ops.attention_fwd(qu.view(B, Tq, H, dk), qv.view(B, Tq, H, dk) if relpos else None,
                                      k.view(B, Tk, H, dk), v.view(B, Tk, H, dk), pproj, m, 1.0 / math.sqrt(dk),
                                      precise=_state["precise"], drop_p=pa, seed=sa, seed_dev=sda)
        y = torch.empty(B, Tq, D, dtype=torch.float32, device=q.device)
        _gemm_nt(ctxv, wo, B * Tq, D, D, y, bias=bo)
        qa_s = _A(qa)
        ctx.save_for_backward(qa_s, qa_s if ka is qa else _A(ka), _A(pe), m, wq, wk, wv, wo, wpos, _A(qu), _A(qv), _A(k), _A(v),
                              _A(pproj), _A(ctxv), lse)
        ctx.meta = (H, pa, sa, sda, same_kv, relpos, bq is not None)
        return y

    @staticmethod
    @_bwd_mode
    def backward(ctx, dy):
        qa, ka, pe, m, wq, wk, wv, wo, wpos, qu, qv, k, v, pproj, ctxv, lse = ctx.saved_tensors
        H, pa, sa, sda, same_kv, relpos, has_b = ctx.meta
        B, Tq, D = qa.shape
        Tk = ka.shape[1]
        dk = D // H
        T = act_dtype()
        g = _to_act(dy)
        dbo = _bgrad(g, B * Tq, D)
        dctx = torch.empty(B, Tq, D, dtype=T, device=g.device)
        with ops.paired():
            dwo = _wgrad(g, ctxv, B * Tq, D, D)
            _gemm_nn(g, wo, B * Tq, D, D, dctx)
        dqu, dqv, dk_, dv_, dpos = ops.attention_bwd(
            qu.view(B, Tq, H, dk), qv.view(B, Tq, H, dk) if relpos else None, k.view(B, Tk, H, dk),
            v.view(B, Tk, H, dk), pproj, m, ctxv, lse, dctx, 1.0 / math.sqrt(dk), precise=_state["precise"],
            drop_p=pa, seed=sa, seed_dev=sda)
        du = dv_bias = dwpos = None
        if relpos:
            dq = torch.empty(B * Tq, D, dtype=T, device=g.device)
            du = torch.zeros(D, dtype=torch.float32, device=g.device)
            dv_bias = torch.zeros(D, dtype=torch.float32, device=g.device)
            ops.head_bias_bwd(dqu, dqv, dq, D, du, dv_bias, B * Tq, D)
            du, dv_bias = du.view(H, dk), dv_bias.view(H, dk)
            dwpos = _wgrad(dpos, pe, pe.shape[0], D, D)
        else:
            dq = dqu.view(B * Tq, D)
        dk2, dv2 = dk_.view(B * Tk, D), dv_.view(B * Tk, D)
        dbq = dbk = dbv = None
        if has_b:
            dbq, dbk, dbv = _bgrad(dq, B * Tq, D), _bgrad(dk2, B * Tk, D), _bgrad(dv2, B * Tk, D)
        dq_in = dkv_in = None
        # each projection: weight gradient + data gradient as one launch (the data gradients chain through `resid`)
        if same_kv:
            t1 = torch.empty(B * Tq, D, dtype=torch.float32, device=g.device)
            with ops.paired():
                dwq = _wgrad(dq, qa, B * Tq, D, D)
                _gemm_nn(dq, wq, B * Tq, D, D, t1)
            t2 = torch.empty_like(t1)
            with ops.paired():
                dwk = _wgrad(dk2, ka, B * Tk, D, D)
                _gemm_nn(dk2, wk, B * Tk, D, D, t2, resid=t1, ldr=D)
            dq_in = torch.empty(B, Tq, D, dtype=torch.float32, device=g.device)
            with ops.paired():
                dwv = _wgrad(dv2, ka, B * Tk, D, D)
                _gemm_nn(dv2, wv, B * Tk, D, D, dq_in, resid=t2, ldr=D)
        else:
            with ops.paired():
                dwq = _wgrad(dq, qa, B * Tq, D, D)
                if ctx.needs_input_grad[0]:
                    dq_in = torch.empty(B, Tq, D, dtype=torch.float32, device=g.device)
                    _gemm_nn(dq, wq, B * Tq, D, D, dq_in)
            need_kv = ctx.needs_input_grad[1]
            t2 = torch.empty(B * Tk, D, dtype=torch.float32, device=g.device) if need_kv else None
            with ops.paired():
                dwk = _wgrad(dk2, ka, B * Tk, D, D)
                if need_kv:
                    _gemm_nn(dk2, wk, B * Tk, D, D, t2)
            with ops.paired():
                dwv = _wgrad(dv2, ka, B * Tk, D, D)
                if need_kv:
                    dkv_in = torch.empty(B, Tk, D, dtype=torch.float32, device=g.device)
                    _gemm_nn(dv2, wv, B * Tk, D, D, dkv_in, resid=t2, ldr=D)
        return (dq_in, dkv_in, None, None, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dwpos, du, dv_bias, None, None,
                None)


def attention_core(q_in, kv_in, pos_emb, mask, wq, bq, wk, bk, wv, bv, wo, bo, wpos, bias_u, bias_v, H, p_attn):
    _state["tag_ok"] = torch.is_grad_enabled()
    same = kv_in is q_in
    return AttentionCoreFn.apply(q_in, q_in if same else kv_in, pos_emb, mask, wq, bq, wk, bk, wv, bv, wo, bo, wpos,
                                 bias_u, bias_v, H, float(p_attn), same)


class MhaSublayerFn(torch.autograd.Function):
    """x + dropout(MHA(LN(x), kv, kv)):  conformer_encoder.py:119-142 (rel-pos self attention, kv = LN(x)) and
    transformer_decoder.py:65-118 (self attention with kv = LN(x); source attention with kv = memory)."""

    @staticmethod
    def forward(ctx, x, memory, pos_emb, mask, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, wpos, bias_u, bias_v, H,
                p_attn, p_out, eps, kv_all=None, kv_slot=0, kv_holder=None, pp_all=None, pp_slot=0, pp_holder=None):
        x = x.contiguous()
        ctx.chain = _chain_take(x)
        B, Tq, D = x.shape
        dk = D // H
        T = act_dtype()
        h, mean, rstd = ops.layernorm_fwd(x, ln_w, ln_b, T, eps, twin=True)
        shared_kv = kv_all is not None  # source attention on the all-layer K/V projection of the memory (MemoryKVFn)
        cross = memory is not None or shared_kv
        if shared_kv:
            ka = None
            Tk = kv_all.shape[0] // B
        else:
            ka = _to_act_shared(memory) if cross else h
            Tk = ka.shape[1]
        # self attention in bf16: ONE projection GEMM onto the concatenated [Wq; Wk; Wv] (N = 3D fills the chip where
        # three N = D launches do not); q / k / v are column thirds of its output, read in place by the attention kernel
        fused = AF._FUSE_QKV and (not cross) and (not _state["precise"]) and D % 64 == 0 and T in (torch.bfloat16, torch.float16) \
            and all(w.dtype == torch.float32 and w.is_contiguous() for w in (wq, wk, wv))
        relpos = pos_emb is not None
        qkv = None
        if fused:
            qkv = torch.empty(B * Tq, 3 * D, dtype=T, device=x.device)
            if T == torch.float16:
                _h16_nt(h, D, _w_h16_cat((wq, wk, wv)), B * Tq, 3 * D, D, qkv, 3 * D, bias=_bias3(bq, bk, bv), twin=True)
            else:
                ops.gemm_bf16_nt(h, D, _w_bf16_cat((wq, wk, wv), False), D, B * Tq, 3 * D, D, qkv, 3 * D,
                                 bias=_bias3(bq, bk, bv))
            q5 = qkv.view(B, Tq, 3, H, dk)
            q, k4, v4 = qkv, q5[:, :, 1], q5[:, :, 2]
            ldq = 3 * D
        elif shared_kv:
            q = _proj(h, wq, bq, B * Tq, D, twin=pos_emb is None)
            assert kv_all.dtype == T, "shared K / V projection and this sub-layer must run the same forward format"
            kv5 = kv_all.view(B, Tk, kv_all.shape[1] // D, H, dk)  # [.., 2 * slot] = K, [.., 2 * slot + 1] = V of this layer
            k4, v4 = kv5[:, :, 2 * kv_slot], kv5[:, :, 2 * kv_slot + 1]
            ldq = D
        else:
            q = _proj(h, wq, bq, B * Tq, D, twin=pos_emb is None)
            k4 = _proj(ka, wk, bk, B * Tk, D).view(B, Tk, H, dk)
            v4 = _proj(ka, wv, bv, B * Tk, D).view(B, Tk, H, dk)
            ldq = D
        pe = pproj = qv = None
        if relpos:
            pe = _to_act_shared(pos_emb).reshape(-1, D)
            pre = _pos_proj.get((pos_emb.data_ptr(), wpos.data_ptr())) if pp_all is None else None
            if pp_all is not None:
                pproj = pp_all[:, pp_slot * D:(pp_slot + 1) * D]  # column block of the all-layer projection (PosProjFn)
            elif pre is not None and pre[1] is not None and pre[1].dtype == T:
                pproj = pre[1]  # the same, without autograd (prepare_pos_proj under no_grad), row pitch n_layers * D
            else:
                pproj = torch.empty(pe.shape[0], D, dtype=T, device=x.device)
                _gemm_nt(pe, wpos, pe.shape[0], D, D, pproj)
            qu, qv = ops.head_bias_fwd(q, ldq, B * Tq, D, bias_u.reshape(-1), bias_v.reshape(-1))
            qu, qv = qu.view(B, Tq, H, dk), qv.view(B, Tq, H, dk)
        else:
            qu = q5[:, :, 0] if fused else q.view(B, Tq, H, dk)
        pa, sa, sda = _drop_args(p_attn, x)
        m = _mask_arg(mask)
        ctxv, lse = ops.attention_fwd(qu, qv, k4, v4, pproj, m, 1.0 / math.sqrt(dk),
                                      precise=_state["precise"], drop_p=pa, seed=sa, seed_dev=sda)
        po, so, sdo = _drop_args(p_out, x)
        y = torch.empty_like(x)
        _gemm_nt(ctxv, wo, B * Tq, D, D, y, bias=bo, drop_p=po, seed=so, seed_dev=sdo, resid=x, ldr=D)
        if fused and qkv.dtype == torch.float16:  # thirds of the fused projection: the same views of its bf16 twin
            s_qu, s_k, s_v = (_A(qu) if relpos else _A_view(qu, qkv)), _A_view(k4, qkv), _A_view(v4, qkv)
        elif shared_kv and kv_all.dtype == torch.float16:
            s_qu, s_k, s_v = _A(qu), _A_view(k4, kv_all), _A_view(v4, kv_all)
        else:
            s_qu, s_k, s_v = _A(qu), _A(k4), _A(v4)
        s_pp = _A_view(pproj, pp_all) if (pp_all is not None and pproj is not None) else _A(pproj)
        ctx.save_for_backward(x, ln_w, mean, rstd, _A(h), _A_shared(ka) if (cross and not shared_kv) else None, _A_shared(pe), m,
                              wq, wk, wv, wo, wpos, s_qu, _A(qv), s_k, s_v, s_pp, _A(ctxv), lse)
        ctx.meta = (H, pa, sa, sda, po, so, sdo, cross, relpos, fused)
        ctx.kv = (kv_slot, kv_holder, tuple(kv_all.shape), kv_all.dtype) if shared_kv else None
        ctx.pp = (pp_slot, pp_holder, tuple(pp_all.shape), pp_all.dtype) if (relpos and pp_all is not None) else None
        _chain_tag(y, B * Tq, D, 1.0, (po, so, sdo))
        return y

    @staticmethod
    @_bwd_mode
    def backward(ctx, dy):
        (x, ln_w, mean, rstd, h, ka, pe, m, wq, wk, wv, wo, wpos, qu, qv, k4, v4, pproj, ctxv, lse) = ctx.saved_tensors
        H, pa, sa, sda, po, so, sdo, cross, relpos, fused = ctx.meta
        dy = dy.contiguous()
        B, Tq, D = x.shape
        shared_kv = ctx.kv is not None
        if not cross:
            ka = h
        Tk = k4.shape[1]
        dk = D // H
        T = act_dtype()
        g, gT, _ = _prologue(dy, B * Tq, D, drop=(po, so, sdo), want_bias=False)
        dbo = _zeros(D, x.device)
        dctx = torch.empty(B, Tq, D, dtype=T, device=x.device)
        with ops.paired():
            dwo = _wgrad(g, ctxv, B * Tq, D, D, bias_out=dbo)
            _gemm_nn(g, wo, B * Tq, D, D, dctx)
        outs = {}
        if fused:  # dq | dk | dv land side by side: one bias-gradient pass, one weight-gradient GEMM, one data-gradient GEMM
            dqkv = torch.empty(B * Tq, 3 * D, dtype=T, device=x.device)
            d5 = dqkv.view(B, Tq, 3, H, dk)
            outs = dict(dk_out=d5[:, :, 1], dv_out=d5[:, :, 2])
            if not relpos:
                outs["dqu_out"] = d5[:, :, 0]
        dkv_grad = None
        if shared_kv:
            # dK / dV go straight into this layer's columns of the shared gradient buffer; the projection's own backward
            # (weight, bias and memory gradients of ALL layers) runs once, in MemoryKVFn.backward
            slot, holder, shape, kv_dtype = ctx.kv
            if holder.get("dkv") is None:
                holder["dkv"] = torch.empty(shape, dtype=T, device=x.device)
            g5 = holder["dkv"].view(B, Tk, shape[1] // D, H, dk)
            outs = dict(dk_out=g5[:, :, 2 * slot], dv_out=g5[:, :, 2 * slot + 1])
            holder["filled"] = holder.get("filled", 0) + 1
            # ONE consumer hands autograd a gradient (the others None); the real buffer travels in `holder` -- autograd would
            # convert it to the dtype of the forward output (f16 in the mixed mode)
            dkv_grad = (holder["dkv"] if kv_dtype == T else _placeholder_grad(shape, kv_dtype, x.device)) if slot == 0 else None
        dpp_grad = None
        if ctx.pp is not None:
            # this layer's position gradient accumulates into its column block of ONE zero-filled buffer; the weight
            # gradients of all layers come from it in PosProjFn.backward
            slot, holder, shape, pp_dtype = ctx.pp
            if holder.get("dpos") is None:
                holder["dpos"] = _zeros(shape, x.device)
            outs = dict(outs, dpos_out=holder["dpos"][:, slot * D:(slot + 1) * D])
            holder["filled"] = holder.get("filled", 0) + 1
            # the f32 buffer travels in `holder`; the placeholder carries the dtype of the forward output (f16 in the mixed mode)
            dpp_grad = _placeholder_grad(shape, pp_dtype, x.device) if slot == 0 else None
        du = dv_bias = dwpos = None
        if relpos:
            # the attention backward itself emits dq = dqu + dqv and the two position-bias gradients (their column sums)
            dq = dqkv if fused else torch.empty(B * Tq, D, dtype=T, device=x.device)
            du = _zeros(D, x.device)
            dv_bias = _zeros(D, x.device)
            if not _state.get("det", False):
                outs = dict(outs, dq_sum=d5[:, :, 0] if fused else dq.view(B, Tq, H, dk), du=du, dv_bias=dv_bias)
        dqu, dqv, dk_, dv_, dpos = ops.attention_bwd(
            qu, qv, k4, v4, pproj, m, ctxv, lse, dctx, 1.0 / math.sqrt(dk), precise=_state["precise"],
            drop_p=pa, seed=sa, seed_dev=sda, **outs)
        if relpos and _state.get("det", False):
            # deterministic mode: the kernel leaves dqu / dqv; ONE ordered pass (a single block per column group) forms
            # dq = dqu + dqv -- straight into the q third of the fused d(qkv) buffer -- and the two position-bias gradients
            ops.head_bias_bwd(dqu.view(B * Tq, D), dqv.view(B * Tq, D), dq, 3 * D if fused else D, du, dv_bias, B * Tq, D)
        if relpos:
            du, dv_bias = du.view(H, dk), dv_bias.view(H, dk)
            dwpos = _wgrad(dpos, pe, pe.shape[0], D, D) if ctx.pp is None else None
        elif not fused:
            dq = dqu.view(B * Tq, D)
        dmem = None
        if fused:
            dbc = _zeros(3 * D, x.device)
            dh = torch.empty(B * Tq, D, dtype=torch.float32, device=x.device)
            wcT = _w_bf16_cat((wq, wk, wv), True)
            with ops.paired():
                dwc = _wgrad(dqkv, h, B * Tq, 3 * D, D, bias_out=dbc)
                ops.gemm_bf16_nt(dqkv, 3 * D, wcT, 3 * D, B * Tq, D, 3 * D, dh, D)
            dwq, dwk, dwv = dwc[:D], dwc[D:2 * D], dwc[2 * D:]
            dbq, dbk, dbv = dbc[:D], dbc[D:2 * D], dbc[2 * D:]
        else:
            if not shared_kv:
                dk2, dv2 = dk_.view(B * Tk, D), dv_.view(B * Tk, D)
            dbq, dbk, dbv = _zeros(D, x.device), _zeros(D, x.device), _zeros(D, x.device)
            # each projection: weight gradient + data gradient as one launch (data gradients chain through `resid`)
            if shared_kv:
                dh = torch.empty(B * Tq, D, dtype=T, device=x.device)
                with ops.paired():
                    dwq = _wgrad(dq, h, B * Tq, D, D, bias_out=dbq)
                    _gemm_nn(dq, wq, B * Tq, D, D, dh)
                dwk = dwv = dbk = dbv = None
            elif cross:
                dh = torch.empty(B * Tq, D, dtype=T, device=x.device)
                with ops.paired():
                    dwq = _wgrad(dq, h, B * Tq, D, D, bias_out=dbq)
                    _gemm_nn(dq, wq, B * Tq, D, D, dh)
                need_mem = ctx.needs_input_grad[1]
                t2 = torch.empty(B * Tk, D, dtype=torch.float32, device=x.device) if need_mem else None
                with ops.paired():
                    dwk = _wgrad(dk2, ka, B * Tk, D, D, bias_out=dbk)
                    if need_mem:
                        _gemm_nn(dk2, wk, B * Tk, D, D, t2)
                with ops.paired():
                    dwv = _wgrad(dv2, ka, B * Tk, D, D, bias_out=dbv)
                    if need_mem:
                        dmem = torch.empty(B, Tk, D, dtype=torch.float32, device=x.device)
                        _gemm_nn(dv2, wv, B * Tk, D, D, dmem, resid=t2, ldr=D)
            else:
                t1 = torch.empty(B * Tq, D, dtype=torch.float32, device=x.device)
                with ops.paired():
                    dwq = _wgrad(dq, h, B * Tq, D, D, bias_out=dbq)
                    _gemm_nn(dq, wq, B * Tq, D, D, t1)
                t2 = torch.empty_like(t1)
                with ops.paired():
                    dwk = _wgrad(dk2, ka, B * Tk, D, D, bias_out=dbk)
                    _gemm_nn(dk2, wk, B * Tq, D, D, t2, resid=t1, ldr=D)
                dh = torch.empty_like(t1)
                with ops.paired():
                    dwv = _wgrad(dv2, ka, B * Tk, D, D, bias_out=dbv)
                    _gemm_nn(dv2, wv, B * Tq, D, D, dh, resid=t2, ldr=D)
        dg = _zeros(D, x.device)
        dbt = _zeros(D, x.device)
        dx = _ln_bwd(dh, x, ln_w, mean, rstd, dg, dbt, dy, ctx.chain)
        return (dx, dmem, None, None, dg, dbt, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dwpos, du, dv_bias, None, None,
                None, None, dkv_grad, None, None, dpp_grad, None, None)


def mha_sublayer(x, memory, pos_emb, mask, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, wpos, bias_u, bias_v, H, p_attn,
                 p_out, eps=1e-12, kv=None):
    """kv = (kv_all, slot, holder) from memory_kv(): source attention reads its K / V from the all-layer projection."""
    _state["tag_ok"] = torch.is_grad_enabled()
    if kv is not None:
        return MhaSublayerFn.apply(_to_f32(x), None, pos_emb, mask, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, wpos,
                                   bias_u, bias_v, H, float(p_attn), float(p_out), eps, kv[0], kv[1], kv[2])
    if pos_emb is not None and wpos is not None:
        pre = _pos_proj.get((pos_emb.data_ptr(), wpos.data_ptr()))
        if pre is not None and pre[2] is not None and torch.is_grad_enabled():
            return MhaSublayerFn.apply(_to_f32(x), memory, pos_emb, mask, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, wpos,
                                       bias_u, bias_v, H, float(p_attn), float(p_out), eps, None, 0, None, pre[2], pre[3],
                                       pre[4])
    return MhaSublayerFn.apply(_to_f32(x), memory, pos_emb, mask, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, wpos,
                               bias_u, bias_v, H, float(p_attn), float(p_out), eps)


class MemoryKVFn(torch.autograd.Function):
    """K and V projections of the encoder memory for ALL decoder layers at once (transformer_decoder.py:100-108 runs
    linear_k / linear_v of every layer's src_attn on the same memory, attention.py:50-52):
        forward : kv_all [B*Tk, 2*n*D] = memory @ [Wk_0; Wv_0; Wk_1; ...]^T + [bk_0 | bv_0 | ...]   -- ONE GEMM instead of 2n;
        backward: the n source-attention sub-layers write dK_l / dV_l into their columns of one shared buffer
                  (MhaSublayerFn, shared_kv); when all of them have run, ONE paired launch gives the weight gradients of
                  all 2n projections (+ bias gradients) and the memory gradient sum_l (dK_l Wk_l + dV_l Wv_l) -- instead of
                  2n paired launches chained through `resid` and n - 1 autograd additions of [B, Tk, D] tensors.
    The contraction of the memory gradient runs over K = 2*n*D = 9216 in one launch: long k loops are where the tile kernel
    is efficient."""

    @staticmethod
    def forward(ctx, memory, holder, *wb):
        B, Tk, D = memory.shape
        ws, bs = wb[0::2], wb[1::2]
        n = len(ws)
        ma = _to_act_shared(memory).reshape(B * Tk, D)
        kv = torch.empty(B * Tk, n * D, dtype=ma.dtype, device=memory.device)
        if ma.dtype == torch.float16:  # mixed mode: f16 projection + bf16 twin (the source-attention backward passes read views of it)
            _h16_nt(ma, D, _w_h16_cat(tuple(ws)), B * Tk, n * D, D, kv, n * D, bias=torch.cat(bs), twin=True)
        else:
            ops.gemm_bf16_nt(ma, D, _w_bf16_cat(tuple(ws), False), D, B * Tk, n * D, D, kv, n * D, bias=torch.cat(bs))
        ctx.save_for_backward(_A_shared(ma), *ws)
        ctx.holder = holder
        ctx.meta = (B, Tk, D, n)
        return kv

    @staticmethod
    @_bwd_mode
    def backward(ctx, dkv):
        ma, *ws = ctx.saved_tensors
        B, Tk, D, n = ctx.meta
        holder = ctx.holder
        assert holder.get("filled", 0) == n // 2 and holder.get("dkv") is not None, \
            "MemoryKVFn: every source-attention sub-layer must have written its dK / dV"
        dkv = holder["dkv"]  # (the autograd-visible gradient is a placeholder of the forward dtype)
        rows = B * Tk
        dbias = _zeros(n * D, dkv.device)
        dmem = torch.empty(B, Tk, D, dtype=torch.float32, device=dkv.device)
        wcT = _w_bf16_cat(tuple(ws), True)
        with ops.paired():
            dW = _wgrad(dkv, ma, rows, n * D, D, bias_out=dbias)
            ops.gemm_bf16_nt(dkv, n * D, wcT, n * D, rows, D, n * D, dmem.view(rows, D), D)
        holder["dkv"] = None
        holder["filled"] = 0
        grads = [dmem, None]
        for i in range(n):
            grads += [dW[i * D:(i + 1) * D], dbias[i * D:(i + 1) * D]]
        return tuple(grads)


def memory_kv(memory, layers_kv):
    """layers_kv: [(Wk, bk, Wv, bv)] per decoder layer.  Returns (kv_all, holder) for mha_sublayer(kv=(kv_all, l, holder)),
    or None when the shared projection does not apply (precise mode, shapes the tuned kernel does not take)."""
    if _state["precise"] or not AF._FUSE_QKV or len(layers_kv) < 2 or not torch.is_grad_enabled():
        return None
    D = memory.shape[-1]
    flat = []
    for (wk, bk, wv, bv) in layers_kv:
        flat += [wk, bk, wv, bv]
    if D % 64 or any(w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != (D, D) for w in flat[0::2]) \
            or any(b is None for b in flat[1::2]) or not memory.requires_grad:
        return None
    holder = {}
    return MemoryKVFn.apply(memory, holder, *flat), holder
