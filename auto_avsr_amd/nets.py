"""Host-side mirror of the reference's ``espnet.nets.pytorch_backend`` module surface for the hot path.

Class names, constructor signatures, attribute names and ``state_dict`` keys follow the reference so that
checkpoints and ``lightning.ModelModule`` keep working; the forward/backward arithmetic is
delegated to the HIP kernels through ``auto_avsr_amd.functional``.  The ``espnet/`` package at the repository
root re-exports these classes under the reference's import paths.

Unsupported corners of the reference API (never reached from train.py / eval.py) raise NotImplementedError
instead of silently running elsewhere: post-norm / concat_after variants, ``zero_triu``, ``rtn_attn``.
"""
import math

import torch
from torch import nn

from . import functional as AF
from .scorer_interface import BatchScorerInterface


# ================================================================================================ building blocks
class LayerNorm(nn.LayerNorm):
    """transformer/layer_norm.py:12-33: LayerNorm over `dim` with eps = 1e-12."""

    def __init__(self, nout, dim=-1):
        super().__init__(nout, eps=1e-12)
        self.dim = dim

    def forward(self, x):
        if self.dim == -1:
            return AF.layer_norm(x, self.weight, self.bias, self.eps)
        return AF.layer_norm(x.transpose(1, -1), self.weight, self.bias, self.eps).transpose(1, -1)


class PositionwiseFeedForward(nn.Module):
    """transformer/positionwise_feed_forward.py:12-30: w_2(dropout(relu(w_1(x))))."""

    def __init__(self, idim, hidden_units, dropout_rate):
        super().__init__()
        self.w_1 = nn.Linear(idim, hidden_units)
        self.w_2 = nn.Linear(hidden_units, idim)
        self.dropout = nn.Dropout(dropout_rate)

    def forward(self, x):
        p = self.dropout.p if self.training else 0.0
        return AF.ffn(x, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias, p)


class MultiHeadedAttention(nn.Module):
    """transformer/attention.py:16-104.  ``self.attn`` (the materialised probabilities of the reference) is never
    formed by the fused kernel and stays None."""

    def __init__(self, n_head, n_feat, dropout_rate):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k = n_feat // n_head
        self.h = n_head
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.attn = None
        self.dropout = nn.Dropout(p=dropout_rate)
        self._pack_qkv_bias()

    def _pack_qkv_bias(self):
        """Keep the three projection biases back to back in ONE buffer (each nn.Parameter is a view of its third): the
        fused Q/K/V projection then reads its [3D] bias in place, without a per-step concatenation launch.  Names, shapes
        and state_dict entries are untouched; if the layout is ever lost (e.g. copy.deepcopy) functional._bias3 simply
        concatenates again."""
        bq, bk, bv = self.linear_q.bias, self.linear_k.bias, self.linear_v.bias
        n = bq.numel()
        if bq.dtype != torch.float32 or bq.device.type == "meta":
            return
        if bk.data_ptr() == bq.data_ptr() + 4 * n and bv.data_ptr() == bq.data_ptr() + 8 * n \
                and bq.untyped_storage().data_ptr() == bv.untyped_storage().data_ptr():
            return  # (adjacent addresses alone do not do: three separate allocations often ARE back to back on the device)
        with torch.no_grad():
            buf = torch.cat([bq.data, bk.data, bv.data])
            bq.data, bk.data, bv.data = buf[:n], buf[n:2 * n], buf[2 * n:]

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)  # .to() / .cuda() give every parameter a storage of its own again
        self._pack_qkv_bias()
        return out

    def _params(self):
        return (self.linear_q.weight, self.linear_q.bias, self.linear_k.weight, self.linear_k.bias,
                self.linear_v.weight, self.linear_v.bias, self.linear_out.weight, self.linear_out.bias)

    def _p_attn(self):
        return self.dropout.p if self.training else 0.0

    def forward(self, query, key, value, mask, rtn_attn=False):
        if rtn_attn:
            raise NotImplementedError("rtn_attn: the fused kernel does not materialise the attention matrix")
        if key is not value:
            raise NotImplementedError("key and value must be the same tensor (as in every call site of the reference)")
        return AF.attention_core(query, key, None, mask, *self._params(), None, None, None, self.h, self._p_attn())


class RelPositionMultiHeadedAttention(MultiHeadedAttention):
    """transformer/attention.py:107-193 (Transformer-XL style relative positions, new espnet implementation)."""

    def __init__(self, n_head, n_feat, dropout_rate, zero_triu=False):
        super().__init__(n_head, n_feat, dropout_rate)
        if zero_triu:
            raise NotImplementedError("zero_triu=True is not used by the reference model")
        self.zero_triu = zero_triu
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.empty(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.empty(self.h, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)

    def forward(self, query, key, value, pos_emb, mask):
        if not (query is key and key is value):
            raise NotImplementedError("relative-position attention is self-attention (query is key is value)")
        return AF.attention_core(query, query, pos_emb, mask, *self._params(), self.linear_pos.weight,
                                 self.pos_bias_u, self.pos_bias_v, self.h, self._p_attn())


def _sinusoid(positions, d_model):
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    ang = positions.to(torch.float32).unsqueeze(1) * div
    pe = torch.zeros(positions.numel(), d_model)
    pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


def _drop_legacy_pe(state_dict, prefix, *_):
    state_dict.pop(prefix + "pe", None)  # embedding.py:14-30: old checkpoints stored the table


class PositionalEncoding(nn.Module):
    """transformer/embedding.py:33-87: x*sqrt(d) + sin/cos table, dropout.  The table is a plain attribute (not a
    buffer), exactly like the reference, so it never appears in ``state_dict``."""

    def __init__(self, d_model, dropout_rate, max_len=5000, reverse=False):
        super().__init__()
        self.d_model = d_model
        self.reverse = reverse
        self.xscale = math.sqrt(d_model)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.pe = None
        self.extend_pe(torch.tensor(0.0).expand(1, max_len))
        self._register_load_state_dict_pre_hook(_drop_legacy_pe)

    def extend_pe(self, x):
        n = x.size(1)
        if self.pe is not None and self.pe.size(1) >= n:
            if self.pe.device != x.device:
                self.pe = self.pe.to(x.device)
            return
        pos = torch.arange(n - 1, -1, -1) if self.reverse else torch.arange(n)
        self.pe = _sinusoid(pos, self.d_model).unsqueeze(0).to(x.device)

    def table(self, n, device):
        self.extend_pe(torch.empty(1, n, device=device))
        return self.pe[0, :n]

    def forward(self, x):
        n = x.size(1)
        return AF.AddRowsFn.apply(x, self.table(n, x.device).contiguous(), self.xscale,
                                  self.dropout.p if self.training else 0.0)


class ScaledPositionalEncoding(PositionalEncoding):
    """transformer/embedding.py:90-117 -- present for import compatibility; not used by the reference model."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__(d_model=d_model, dropout_rate=dropout_rate, max_len=max_len)
        self.alpha = nn.Parameter(torch.tensor(1.0))

    def reset_parameters(self):
        self.alpha.data = torch.tensor(1.0)

    def forward(self, x):
        raise NotImplementedError("ScaledPositionalEncoding is unused by auto_avsr's E2E (SURVEY.md section 2)")


class RelPositionalEncoding(nn.Module):
    """transformer/embedding.py:120-184: returns (dropout(x*sqrt(d)), dropout(pos_emb)); pos_emb row k is the
    sinusoid of relative position T-1-k, k = 0..2T-2."""

    def __init__(self, d_model, dropout_rate, max_len=5000):
        super().__init__()
        self.d_model = d_model
        self.xscale = math.sqrt(d_model)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.pe = None
        self.extend_pe(torch.tensor(0.0).expand(1, max_len))

    def extend_pe(self, x):
        n = x.size(1)
        if self.pe is not None and self.pe.size(1) >= 2 * n - 1:
            if self.pe.device != x.device:
                self.pe = self.pe.to(x.device)
            return
        self.pe = _sinusoid(torch.arange(n - 1, -n, -1), self.d_model).unsqueeze(0).to(x.device)

    def forward(self, x):
        self.extend_pe(x)
        n = x.size(1)
        mid = self.pe.size(1) // 2
        pos_emb = self.pe[:, mid - n + 1: mid + n]
        p = self.dropout.p if self.training else 0.0
        return AF.scale_dropout(x, self.xscale, p), AF.scale_dropout(pos_emb.contiguous(), 1.0, p)


class MultiSequential(nn.Sequential):
    """transformer/repeat.py:8-27: sequential container passing tuples through, with stochastic layer drop."""

    def __init__(self, *args, layer_drop_rate=0.0):
        super().__init__(*args)
        self.layer_drop_rate = layer_drop_rate

    def forward(self, *args):
        probs = torch.empty(len(self)).uniform_() if (self.training and self.layer_drop_rate > 0) else None
        for idx, m in enumerate(self):
            if probs is None or probs[idx] >= self.layer_drop_rate:
                args = m(*args)
        return args


def repeat(N, fn, layer_drop_rate=0.0):
    """transformer/repeat.py:30-42."""
    return MultiSequential(*[fn(n) for n in range(N)], layer_drop_rate=layer_drop_rate)


# ================================================================================================ Conformer encoder
class ConvolutionModule(nn.Module):
    """encoder/conformer_encoder.py:19-35 (parameter names keep the reference's spelling ``pointwise_cov``)."""

    def __init__(self, channels, kernel_size, bias=True):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0
        if not bias:
            raise NotImplementedError("bias=False is not used by the reference model")
        self.pointwise_cov1 = nn.Conv1d(channels, 2 * channels, 1, bias=bias)
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, padding=(kernel_size - 1) // 2,
                                        groups=channels, bias=bias)
        self.norm = nn.BatchNorm1d(channels)
        self.pointwise_cov2 = nn.Conv1d(channels, channels, 1, bias=bias)
        self.activation = nn.SiLU(inplace=True)

    def _params(self):
        return (self.pointwise_cov1.weight, self.pointwise_cov1.bias, self.depthwise_conv.weight,
                self.depthwise_conv.bias, self.norm, self.pointwise_cov2.weight, self.pointwise_cov2.bias)

    def forward(self, x):
        return AF.conv_sublayer(x, None, None, *self._params(), 0.0)


class EncoderLayer(nn.Module):
    """encoder/conformer_encoder.py:38-170: macaron FFN -> rel-pos MHA -> conv module -> FFN -> final LayerNorm."""

    def __init__(self, size, self_attn, feed_forward, conv_module, dropout_rate, normalize_before=True,
                 concat_after=False, macaron_style=False):
        super().__init__()
        import copy

        if not normalize_before or concat_after:
            raise NotImplementedError("only the pre-norm, non-concat block of the reference model is implemented")
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.ff_scale = 1.0
        self.conv_module = conv_module
        self.macaron_style = macaron_style
        self.norm_ff = LayerNorm(size)
        self.norm_mha = LayerNorm(size)
        if macaron_style:
            self.feed_forward_macaron = copy.deepcopy(feed_forward)
            self.ff_scale = 0.5
            self.norm_ff_macaron = LayerNorm(size)
        if conv_module is not None:
            self.norm_conv = LayerNorm(size)
            self.norm_final = LayerNorm(size)
        self.dropout = nn.Dropout(dropout_rate)
        self.size = size
        self.normalize_before = normalize_before
        self.concat_after = concat_after

    def _ffn(self, x, norm, ff):
        p = self.dropout.p if self.training else 0.0
        return AF.ffn_sublayer(x, norm.weight, norm.bias, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias,
                               self.ff_scale, p, norm.eps)

    def forward(self, x_input, mask, cache=None):
        if cache is not None:
            raise NotImplementedError("incremental encoder cache (dead code in the reference, SURVEY F10)")
        x, pos_emb = x_input if isinstance(x_input, tuple) else (x_input, None)
        p = self.dropout.p if self.training else 0.0
        if self.macaron_style:
            x = self._ffn(x, self.norm_ff_macaron, self.feed_forward_macaron)
        a = self.self_attn
        relpos = pos_emb is not None
        x = AF.mha_sublayer(x, None, pos_emb, mask, self.norm_mha.weight, self.norm_mha.bias, *a._params(),
                            a.linear_pos.weight if relpos else None, a.pos_bias_u if relpos else None,
                            a.pos_bias_v if relpos else None, a.h, a._p_attn(), p, self.norm_mha.eps)
        if self.conv_module is not None:
            x = AF.conv_sublayer(x, self.norm_conv.weight, self.norm_conv.bias, *self.conv_module._params(), p,
                                 self.norm_conv.eps)
        x = self._ffn(x, self.norm_ff, self.feed_forward)
        if self.conv_module is not None:
            x = self.norm_final(x)
        return ((x, pos_emb), mask) if relpos else (x, mask)


def _rename(state_dict, old, new):
    for k in [k for k in state_dict if k.startswith(old)]:
        state_dict[new + k[len(old):]] = state_dict.pop(k)


def _encoder_legacy_keys(state_dict, prefix, *_):
    _rename(state_dict, prefix + "input_layer.", prefix + "embed.")
    _rename(state_dict, prefix + "norm.", prefix + "after_norm.")


class ConformerEncoder(nn.Module):
    """encoder/conformer_encoder.py:186-303.  ``relu_type`` is accepted and ignored exactly like the reference
    (the Macaron FFN is ReLU)."""

    def __init__(self, attention_dim=768, attention_heads=12, linear_units=3072, num_blocks=12, dropout_rate=0.1,
                 positional_dropout_rate=0.1, attention_dropout_rate=0.0, normalize_before=True, concat_after=False,
                 macaron_style=True, use_cnn_module=True, zero_triu=False, cnn_module_kernel=31, padding_idx=-1,
                 relu_type="swish", layer_drop_rate=0.0):
        super().__init__()
        self._register_load_state_dict_pre_hook(_encoder_legacy_keys)
        self.embed = nn.Sequential(RelPositionalEncoding(attention_dim, positional_dropout_rate))
        self.normalize_before = normalize_before
        self.encoders = repeat(
            num_blocks,
            lambda n: EncoderLayer(
                attention_dim,
                RelPositionMultiHeadedAttention(attention_heads, attention_dim, attention_dropout_rate, zero_triu),
                PositionwiseFeedForward(attention_dim, linear_units, dropout_rate),
                ConvolutionModule(attention_dim, cnn_module_kernel) if use_cnn_module else None,
                dropout_rate, normalize_before, concat_after, macaron_style),
            layer_drop_rate=0.0,
        )
        if normalize_before:
            self.after_norm = LayerNorm(attention_dim)

    def forward(self, xs, masks):
        with AF.component("encoder"):  # (mixed numerical mode: the forward arithmetic of this component, AF.MIXED_POLICY)
            xs = self.embed(xs)
            if isinstance(xs, tuple):
                # the position table is batch-shared and identical for every layer: project it for all layers in ONE GEMM
                # against the concatenated linear_pos weights (attention.py:170 runs linear_pos once per layer)
                AF.prepare_pos_proj(xs[1], [layer.self_attn.linear_pos.weight for layer in self.encoders
                                            if hasattr(layer.self_attn, "linear_pos")])
            xs, masks = self.encoders(xs, masks)
            if isinstance(xs, tuple):
                xs = xs[0]
            if self.normalize_before:
                xs = self.after_norm(xs)
        return xs, masks

    def forward_one_step(self, xs, masks, cache=None):
        raise NotImplementedError("ConformerEncoder.forward_one_step is dead code in the reference (SURVEY F10)")


# ================================================================================================ Transformer decoder
class DecoderLayer(nn.Module):
    """decoder/transformer_decoder.py:22-128 (pre-norm): self-attention, source attention, FFN."""

    def __init__(self, size, self_attn, src_attn, feed_forward, dropout_rate, normalize_before=True,
                 concat_after=False):
        super().__init__()
        if not normalize_before or concat_after:
            raise NotImplementedError("only the pre-norm, non-concat block of the reference model is implemented")
        self.size = size
        self.self_attn = self_attn
        self.src_attn = src_attn
        self.feed_forward = feed_forward
        self.norm1 = LayerNorm(size)
        self.norm2 = LayerNorm(size)
        self.norm3 = LayerNorm(size)
        self.dropout = nn.Dropout(dropout_rate)
        self.normalize_before = normalize_before
        self.concat_after = concat_after

    def forward(self, tgt, tgt_mask, memory, memory_mask, cache=None, kv=None):
        """kv (not in the reference): (kv_all, slot, holder) from functional.memory_kv -- this layer's source-attention K / V
        are columns of the all-layer projection of the memory (TransformerDecoder.forward)."""
        p = self.dropout.p if self.training else 0.0
        sa, ca, ff = self.self_attn, self.src_attn, self.feed_forward
        if cache is None:
            x = AF.mha_sublayer(tgt, None, None, tgt_mask, self.norm1.weight, self.norm1.bias, *sa._params(), None,
                                None, None, sa.h, sa._p_attn(), p, self.norm1.eps)
        else:
            # incremental step (beam search): only the last position is a query; keys/values are all positions
            assert cache.shape == (tgt.shape[0], tgt.shape[1] - 1, self.size)
            h = self.norm1(tgt)
            q_mask = None if tgt_mask is None else tgt_mask[:, -1:, :]
            att = AF.AttentionCoreFn.apply(h[:, -1:, :].contiguous(), h, None, q_mask, *sa._params(), None, None,
                                           None, sa.h, sa._p_attn(), False)
            x = AF.add(tgt[:, -1:, :], att)
        x = AF.mha_sublayer(x, memory, None, memory_mask, self.norm2.weight, self.norm2.bias, *ca._params(), None, None,
                            None, ca.h, ca._p_attn(), p, self.norm2.eps, kv=kv)
        x = AF.ffn_sublayer(x, self.norm3.weight, self.norm3.bias, ff.w_1.weight, ff.w_1.bias, ff.w_2.weight,
                            ff.w_2.bias, 1.0, p, self.norm3.eps)
        if cache is not None:
            x = torch.cat([cache, x], dim=1)
        return x, tgt_mask, memory, memory_mask


def _decoder_legacy_keys(state_dict, prefix, *_):
    _rename(state_dict, prefix + "output_norm.", prefix + "after_norm.")


class TransformerDecoder(BatchScorerInterface, nn.Module):
    """decoder/transformer_decoder.py:144-334 (a BatchScorerInterface like the reference's, :144).  Teacher-forced ``forward`` is the training hot path; the
    incremental scorer API (``forward_one_step`` / ``score`` / ``batch_score``) serves beam search."""

    def __init__(self, odim, attention_dim=256, attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                 positional_dropout_rate=0.1, self_attention_dropout_rate=0.1, src_attention_dropout_rate=0.1,
                 input_layer="embed", use_output_layer=True, pos_enc_class=PositionalEncoding, normalize_before=True,
                 concat_after=False, layer_drop_rate=0.0):
        super().__init__()
        self._register_load_state_dict_pre_hook(_decoder_legacy_keys)
        if input_layer != "embed":
            raise NotImplementedError("only input_layer='embed' (the reference model) is implemented")
        self.embed = nn.Sequential(nn.Embedding(odim, attention_dim),
                                   pos_enc_class(attention_dim, positional_dropout_rate))
        self.normalize_before = normalize_before
        self.decoders = repeat(
            num_blocks,
            lambda n: DecoderLayer(
                attention_dim,
                MultiHeadedAttention(attention_heads, attention_dim, self_attention_dropout_rate),
                MultiHeadedAttention(attention_heads, attention_dim, src_attention_dropout_rate),
                PositionwiseFeedForward(attention_dim, linear_units, dropout_rate),
                dropout_rate, normalize_before, concat_after),
            layer_drop_rate,
        )
        if normalize_before:
            self.after_norm = LayerNorm(attention_dim)
        self.output_layer = nn.Linear(attention_dim, odim) if use_output_layer else None

    def _embed(self, tgt):
        emb, pe = self.embed[0], self.embed[1]
        p = pe.dropout.p if self.training else 0.0
        return AF.embed(tgt, emb.weight, pe.table(tgt.shape[1], emb.weight.device), pe.xscale, p)

    def forward(self, tgt, tgt_mask, memory, memory_mask):
        with AF.component("decoder"):  # (mixed numerical mode: forward arithmetic of this component, AF.MIXED_POLICY)
            x = self._embed(tgt)
            # the source-attention K / V projections of all layers act on the same memory: one GEMM (functional.MemoryKVFn)
            shared = AF.memory_kv(memory, [(d.src_attn.linear_k.weight, d.src_attn.linear_k.bias, d.src_attn.linear_v.weight,
                                            d.src_attn.linear_v.bias) for d in self.decoders])
            if shared is None:
                x, tgt_mask, memory, memory_mask = self.decoders(x, tgt_mask, memory, memory_mask)
            else:
                for i, layer in enumerate(self.decoders):  # (layer_drop_rate is 0.0 in the reference model: plain loop)
                    x, tgt_mask, memory, memory_mask = layer(x, tgt_mask, memory, memory_mask, kv=(shared[0], i, shared[1]))
            if self.normalize_before:
                x = self.after_norm(x)
        if self.output_layer is not None:
            with AF.component("dec_out"):
                x = AF.linear(x, self.output_layer.weight, self.output_layer.bias, out_dtype=torch.float32, pad_out=True)
        return x, tgt_mask

    def forward_one_step(self, tgt, tgt_mask, memory, memory_mask=None, cache=None):
        x = self._embed(tgt)
        if cache is None:
            cache = [None] * len(self.decoders)
        new_cache = []
        for c, dec in zip(cache, self.decoders):
            x, tgt_mask, memory, memory_mask = dec(x, tgt_mask, memory, memory_mask, cache=c)
            new_cache.append(x)
        y = x[:, -1]
        if self.normalize_before:
            y = self.after_norm(y)
        if self.output_layer is not None:
            y = AF.linear(y, self.output_layer.weight, self.output_layer.bias, out_dtype=torch.float32, pad_out=True)
            y = AF.log_softmax(y)
        return y, new_cache

    def score(self, ys, state, x):
        ys_mask = subsequent_mask(len(ys), device=x.device).unsqueeze(0)
        logp, state = self.forward_one_step(ys.unsqueeze(0), ys_mask, x.unsqueeze(0), cache=state)
        return logp.squeeze(0), state

    def batch_score(self, ys, states, xs):
        n_batch, n_layers = len(ys), len(self.decoders)
        if states[0] is None:
            batch_state = None
        else:
            batch_state = [torch.stack([states[b][i] for b in range(n_batch)]) for i in range(n_layers)]
        ys_mask = subsequent_mask(ys.size(-1), device=xs.device).unsqueeze(0)
        logp, states = self.forward_one_step(ys, ys_mask, xs, cache=batch_state)
        return logp, [[states[i][b] for i in range(n_layers)] for b in range(n_batch)]


# ================================================================================================ heads / losses
class CTC(nn.Module):
    """ctc.py:8-93."""

    def __init__(self, odim, eprojs, dropout_rate, reduce=True):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.loss = None
        self.ctc_lo = nn.Linear(eprojs, odim)
        self.dropout = nn.Dropout(dropout_rate)
        self.probs = None
        self.ignore_id = -1
        self.reduce = reduce
        if not reduce:
            raise NotImplementedError("reduce=False is not used by the reference model")

    def _logits(self, hs_pad, train_dropout):
        p = self.dropout.p if (train_dropout and self.training) else 0.0
        h = AF.scale_dropout(hs_pad, 1.0, p, out_dtype=AF.act_dtype())
        return AF.linear(h, self.ctc_lo.weight, self.ctc_lo.bias, out_dtype=torch.float32, pad_out=True)

    def forward(self, hs_pad, hlens, ys_pad):
        logits = self._logits(hs_pad, True)  # (B, T, V)
        self.loss = AF.ctc_loss(logits, ys_pad, hlens.to(logits.device), self.ignore_id)
        return self.loss, logits.transpose(0, 1)

    def log_softmax(self, hs_pad):
        return AF.log_softmax(self._logits(hs_pad, False))

    def softmax(self, hs_pad):
        self.probs = torch.exp(self.log_softmax(hs_pad))
        return self.probs

    def argmax(self, hs_pad):
        return torch.argmax(self._logits(hs_pad, False), dim=-1)


class LabelSmoothingLoss(nn.Module):
    """transformer/label_smoothing_loss.py:14-63 (KL divergence against the smoothed one-hot, summed, / batch)."""

    def __init__(self, size, padding_idx, smoothing, normalize_length=False, criterion=None):
        super().__init__()
        if criterion is not None and not isinstance(criterion, nn.KLDivLoss):
            raise NotImplementedError("only the default KLDivLoss criterion is implemented")
        self.criterion = criterion if criterion is not None else nn.KLDivLoss(reduction="none")
        self.padding_idx = padding_idx
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing
        self.size = size
        self.true_dist = None
        self.normalize_length = normalize_length
        self.last_hits = None  # device scalar: number of correct arg-max predictions of the last call

    def forward(self, x, target):
        assert x.size(2) == self.size
        if self.normalize_length:
            denom = float((target != self.padding_idx).sum().item())
        else:
            denom = float(x.size(0))
        loss, hits = AF.ce_smooth(x, target, self.smoothing, self.padding_idx, denom)
        self.last_hits = hits
        return loss


# ================================================================================================ host glue
def subsequent_mask(size, device="cpu", dtype=torch.bool):
    """transformer/mask.py:11-24."""
    return torch.tril(torch.ones(size, size, device=device, dtype=dtype))


def target_mask(ys_in_pad, ignore_id):
    """transformer/mask.py:27-37."""
    L = ys_in_pad.size(-1)
    return (ys_in_pad != ignore_id).unsqueeze(-2) & subsequent_mask(L, device=ys_in_pad.device).unsqueeze(0)


def pad_list(xs, pad_value):
    """nets_utils.py:34-61."""
    n = len(xs)
    m = max(x.size(0) for x in xs)
    out = xs[0].new_full((n, m) + tuple(xs[0].shape[1:]), pad_value)
    for i, x in enumerate(xs):
        out[i, : x.size(0)] = x
    return out


def add_sos_eos(ys_pad, sos, eos, ignore_id):
    """transformer/add_sos_eos.py:12-31 (data-dependent output width; host sync like the reference)."""
    ys = [y[y != ignore_id] for y in ys_pad]
    s, e = ys_pad.new_tensor([sos]), ys_pad.new_tensor([eos])
    return pad_list([torch.cat([s, y]) for y in ys], eos), pad_list([torch.cat([y, e]) for y in ys], ignore_id)


def add_sos_eos_static(ys_pad, sos, eos, ignore_id):
    """Same targets as add_sos_eos but with the static width Lmax+1 and no host synchronisation: labels are
    compacted to the left on the device.  Extra all-padding columns carry ignore_id in ys_out, so every loss and
    accuracy value is unchanged ("host syncs")."""
    y = ys_pad.reshape(ys_pad.shape[0], -1)
    B, L = y.shape
    keep = y != ignore_id
    n = keep.sum(1, keepdim=True)
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
    comp = torch.gather(y, 1, order)
    ar = torch.arange(L + 1, device=y.device).unsqueeze(0)
    body = torch.cat([comp, comp.new_full((B, 1), eos)], 1)
    ys_out = torch.where(ar < n, body, torch.where(ar == n, eos, ignore_id))  # python scalars: no H2D copies
    shifted = torch.cat([comp.new_full((B, 1), sos), comp], 1)
    ys_in = torch.where(ar <= n, shifted, eos)
    return ys_in, ys_out


def make_pad_mask(lengths, xs=None, length_dim=-1, maxlen=None):
    """nets_utils.py:64-180 (True = padded)."""
    if length_dim == 0:
        raise ValueError("length_dim cannot be 0: {}".format(length_dim))
    if not isinstance(lengths, list):
        lengths = lengths.tolist()
    bs = len(lengths)
    if maxlen is None:
        maxlen = int(max(lengths)) if xs is None else xs.size(length_dim)
    else:
        assert xs is None and maxlen >= int(max(lengths))
    mask = torch.arange(maxlen).unsqueeze(0).expand(bs, maxlen) >= torch.tensor(lengths).unsqueeze(-1)
    if xs is not None:
        assert xs.size(0) == bs, (xs.size(0), bs)
        if length_dim < 0:
            length_dim = xs.dim() + length_dim
        ind = tuple(slice(None) if i in (0, length_dim) else None for i in range(xs.dim()))
        mask = mask[ind].expand_as(xs).to(xs.device)
    return mask


def make_non_pad_mask(lengths, xs=None, length_dim=-1):
    """nets_utils.py:183-269."""
    return ~make_pad_mask(lengths, xs, length_dim)


def non_pad_mask_device(lengths, maxlen):
    """(B,1,T) key-padding mask built on the device of `lengths` (no .tolist() host sync)."""
    return (torch.arange(maxlen, device=lengths.device).unsqueeze(0) < lengths.unsqueeze(1)).unsqueeze(-2)


def th_accuracy(pad_outputs, pad_targets, ignore_label):
    """nets_utils.py:272-292."""
    pred = pad_outputs.view(pad_targets.size(0), pad_targets.size(1), pad_outputs.size(1)).argmax(2)
    mask = pad_targets != ignore_label
    return float(torch.sum(pred.masked_select(mask) == pad_targets.masked_select(mask))) / float(torch.sum(mask))


def to_device(m, x):
    """nets_utils.py:12-31."""
    if isinstance(m, nn.Module):
        return x.to(next(m.parameters()).device)
    if isinstance(m, torch.Tensor):
        return x.to(m.device)
    raise TypeError("Expected torch.nn.Module or torch.tensor, bot got: {}".format(type(m)))


def rename_state_dict(old_prefix, new_prefix, state_dict):
    """nets_utils.py:295-306."""
    _rename(state_dict, old_prefix, new_prefix)
