"""oracle/avsr_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch, fp32, CPU restatement of the reference hot path
``E2E.forward`` (espnet/nets/pytorch_backend/e2e_asr_conformer.py:63-87) written as
pure functions over a ``state_dict`` (reference key names).  It exists only to check
the HIP kernels: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; nothing under ``auto_avsr_amd/`` or ``espnet/`` does.

Pinning: the reference publishes no golden vectors, so the oracle is
pinned against outputs of the reference itself, generated in the build container by
``tests/golden/make_golden.py`` (which imports /root/reference) and committed under
``tests/golden/``; ``tests/test_oracle_golden.py`` replays them.

Every function cites the reference lines it restates.  Dropout is the identity here
(parity runs use p = 0 / eval-mode dropout); BatchNorm uses batch statistics when
``train_bn`` is true -- over every frame including padding, exactly like the reference
 -- and updates nothing.
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12  # transformer/layer_norm.py:21
BN_EPS = 1e-5   # torch default, resnet.py:34 / conformer_encoder.py:26


# --------------------------------------------------------------------------- small pieces
def layer_norm(sd, pre, x):
    """transformer/layer_norm.py:12-33."""
    return F.layer_norm(x, (x.shape[-1],), sd[pre + "weight"], sd[pre + "bias"], LN_EPS)


def linear(sd, pre, x):
    return F.linear(x, sd[pre + "weight"], sd.get(pre + "bias"))


def batch_norm(sd, pre, x, train_bn):
    """nn.BatchNorm{1,2,3}d in train (batch statistics) or eval (running statistics) mode; x is (N, C, ...)."""
    if train_bn:
        dims = [0] + list(range(2, x.dim()))
        mean = x.mean(dims, keepdim=True)
        var = x.var(dims, unbiased=False, keepdim=True)
    else:
        shape = [1, -1] + [1] * (x.dim() - 2)
        mean = sd[pre + "running_mean"].view(shape)
        var = sd[pre + "running_var"].view(shape)
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean) / torch.sqrt(var + BN_EPS) * sd[pre + "weight"].view(shape) + sd[pre + "bias"].view(shape)


def feed_forward(sd, pre, x):
    """transformer/positionwise_feed_forward.py:28-30 -- ReLU, not Swish."""
    return linear(sd, pre + "w_2.", torch.relu(linear(sd, pre + "w_1.", x)))


def sinusoid_table(positions, d_model):
    """embedding.py:67-75: pe[:, 0::2] = sin(pos*div), pe[:, 1::2] = cos(pos*div)."""
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(len(positions), d_model)
    ang = positions.to(torch.float32).unsqueeze(1) * div
    pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


def rel_pos_emb(T, d_model):
    """embedding.py:139-184: row k of the (2T-1, d) table is the sinusoid of relative position T-1-k."""
    rel = torch.arange(T - 1, -T, -1)
    return sinusoid_table(rel, d_model)


def masked_softmax_attention(scores, mask, v):
    """attention.py:59-88: fill finfo.min -> softmax -> zero the masked entries -> @ V."""
    if mask is not None:
        m = mask.unsqueeze(1).eq(0)
        scores = scores.masked_fill(m, torch.finfo(scores.dtype).min)
        attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    else:
        attn = torch.softmax(scores, dim=-1)
    B, H, Tq, _ = attn.shape
    return torch.matmul(attn, v).transpose(1, 2).reshape(B, Tq, -1)


def split_heads(x, H):
    B, T, D = x.shape
    return x.view(B, T, H, D // H).transpose(1, 2)  # (B, H, T, dk)


def mha(sd, pre, q_in, kv_in, mask, H):
    """attention.py:38-57,90-104 (decoder self / source attention)."""
    q = split_heads(linear(sd, pre + "linear_q.", q_in), H)
    k = split_heads(linear(sd, pre + "linear_k.", kv_in), H)
    v = split_heads(linear(sd, pre + "linear_v.", kv_in), H)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(q.shape[-1])
    return linear(sd, pre + "linear_out.", masked_softmax_attention(scores, mask, v))


def rel_mha(sd, pre, x, pos_emb, mask, H):
    """attention.py:153-193; rel_shift (:131-151) restated as the index map bd[i,j] = G[i, j-i+T-1]."""
    B, T, D = x.shape
    q = split_heads(linear(sd, pre + "linear_q.", x), H)
    k = split_heads(linear(sd, pre + "linear_k.", x), H)
    v = split_heads(linear(sd, pre + "linear_v.", x), H)
    p = F.linear(pos_emb, sd[pre + "linear_pos.weight"]).view(2 * T - 1, H, D // H).transpose(0, 1)  # (H, 2T-1, dk)
    qu = q + sd[pre + "pos_bias_u"].view(1, H, 1, -1)
    qv = q + sd[pre + "pos_bias_v"].view(1, H, 1, -1)
    ac = torch.matmul(qu, k.transpose(-2, -1))
    g = torch.matmul(qv, p.transpose(-2, -1).unsqueeze(0))  # (B, H, T, 2T-1)
    i = torch.arange(T).view(T, 1)
    j = torch.arange(T).view(1, T)
    bd = g[:, :, i, j - i + T - 1]
    scores = (ac + bd) / math.sqrt(D // H)
    return linear(sd, pre + "linear_out.", masked_softmax_attention(scores, mask, v))


def conv_module(sd, pre, x, train_bn):
    """conformer_encoder.py:30-35: pw conv -> GLU -> depthwise k conv -> BN -> SiLU -> pw conv (no padding mask)."""
    y = x.transpose(1, 2)
    y = F.glu(F.conv1d(y, sd[pre + "pointwise_cov1.weight"], sd[pre + "pointwise_cov1.bias"]), dim=1)
    w = sd[pre + "depthwise_conv.weight"]
    y = F.conv1d(y, w, sd[pre + "depthwise_conv.bias"], padding=(w.shape[-1] - 1) // 2, groups=w.shape[0])
    y = F.silu(batch_norm(sd, pre + "norm.", y, train_bn))
    y = F.conv1d(y, sd[pre + "pointwise_cov2.weight"], sd[pre + "pointwise_cov2.bias"])
    return y.transpose(1, 2)


def encoder_layer(sd, pre, x, pos_emb, mask, H, train_bn):
    """conformer_encoder.py:96-170 with macaron_style, normalize_before, conv module; ff_scale = 0.5 (:83)."""
    x = x + 0.5 * feed_forward(sd, pre + "feed_forward_macaron.", layer_norm(sd, pre + "norm_ff_macaron.", x))
    x = x + rel_mha(sd, pre + "self_attn.", layer_norm(sd, pre + "norm_mha.", x), pos_emb, mask, H)
    x = x + conv_module(sd, pre + "conv_module.", layer_norm(sd, pre + "norm_conv.", x), train_bn)
    x = x + 0.5 * feed_forward(sd, pre + "feed_forward.", layer_norm(sd, pre + "norm_ff.", x))
    return layer_norm(sd, pre + "norm_final.", x)


def count_layers(sd, pre):
    n = 0
    while any(k.startswith(f"{pre}{n}.") for k in sd):
        n += 1
    return n


def conformer_encoder(sd, pre, x, mask, H, train_bn=True):
    """conformer_encoder.py:264-282 (+ embedding.py:171-184: x*sqrt(d), pos_emb slice)."""
    D = x.shape[-1]
    x = x * math.sqrt(D)
    pos = rel_pos_emb(x.shape[1], D)
    for n in range(count_layers(sd, pre + "encoders.")):
        x = encoder_layer(sd, f"{pre}encoders.{n}.", x, pos, mask, H, train_bn)
    return layer_norm(sd, pre + "after_norm.", x)


def decoder_layer(sd, pre, x, tgt_mask, memory, memory_mask, H):
    """transformer_decoder.py:65-128 (normalize_before, no cache)."""
    h = layer_norm(sd, pre + "norm1.", x)
    x = x + mha(sd, pre + "self_attn.", h, h, tgt_mask, H)
    x = x + mha(sd, pre + "src_attn.", layer_norm(sd, pre + "norm2.", x), memory, memory_mask, H)
    return x + feed_forward(sd, pre + "feed_forward.", layer_norm(sd, pre + "norm3.", x))


def transformer_decoder(sd, pre, ys_in, tgt_mask, memory, memory_mask, H):
    """transformer_decoder.py:229-258 (+ embedding.py:78-87: emb*sqrt(d) + absolute sinusoid)."""
    emb = sd[pre + "embed.0.weight"]
    D = emb.shape[1]
    x = emb[ys_in] * math.sqrt(D) + sinusoid_table(torch.arange(ys_in.shape[1]), D).unsqueeze(0)
    for n in range(count_layers(sd, pre + "decoders.")):
        x = decoder_layer(sd, f"{pre}decoders.{n}.", x, tgt_mask, memory, memory_mask, H)
    return linear(sd, pre + "output_layer.", layer_norm(sd, pre + "after_norm.", x))


# --------------------------------------------------------------------------- visual front-end
def basic_block(sd, pre, x, stride, train_bn):
    """frontend/resnet.py:82-98: conv3x3-bn-silu-conv3x3-bn (+1x1 s conv+bn shortcut) -> add -> silu."""
    out = F.conv2d(x, sd[pre + "conv1.weight"], stride=stride, padding=1)
    out = F.silu(batch_norm(sd, pre + "bn1.", out, train_bn))
    out = F.conv2d(out, sd[pre + "conv2.weight"], padding=1)
    out = batch_norm(sd, pre + "bn2.", out, train_bn)
    if pre + "downsample.0.weight" in sd:
        x = batch_norm(sd, pre + "downsample.1.", F.conv2d(x, sd[pre + "downsample.0.weight"], stride=stride), train_bn)
    return F.silu(out + x)


def video_frontend(sd, pre, x, train_bn=True):
    """frontend/resnet.py:221-233: (B,T,1,88,88) -> Conv3d stem + BN3d + SiLU + MaxPool3d -> ResNet-18 -> (B,T,512)."""
    B, T = x.shape[0], x.shape[1]
    y = x.transpose(1, 2)  # (B,1,T,H,W)
    y = F.conv3d(y, sd[pre + "frontend3D.0.weight"], stride=(1, 2, 2), padding=(2, 3, 3))
    y = F.silu(batch_norm(sd, pre + "frontend3D.1.", y, train_bn))
    y = F.max_pool3d(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    y = y.transpose(1, 2).reshape(B * T, y.shape[1], y.shape[3], y.shape[4])
    for li in range(1, 5):
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            y = basic_block(sd, f"{pre}trunk.layer{li}.{bi}.", y, stride, train_bn)
    return y.mean(dim=(2, 3)).view(B, T, -1)


def basic_block_1d(sd, pre, x, stride, train_bn):
    """frontend/resnet1d.py:83-99."""
    out = F.conv1d(x, sd[pre + "conv1.weight"], stride=stride, padding=1)
    out = F.silu(batch_norm(sd, pre + "bn1.", out, train_bn))
    out = F.conv1d(out, sd[pre + "conv2.weight"], padding=1)
    out = batch_norm(sd, pre + "bn2.", out, train_bn)
    if pre + "downsample.0.weight" in sd:
        x = batch_norm(sd, pre + "downsample.1.", F.conv1d(x, sd[pre + "downsample.0.weight"], stride=stride), train_bn)
    return F.silu(out + x)


def audio_frontend(sd, pre, x, train_bn=True):
    """frontend/resnet1d.py:224-234,188-201: (B,S,1) -> conv k80 s4 + BN + SiLU -> ResNet-18 1-D -> AvgPool(20) -> (B,S/640,512)."""
    B = x.shape[0]
    x = x[:, : x.shape[1] // 640 * 640, :].transpose(1, 2)
    y = F.conv1d(x, sd[pre + "trunk.conv1.weight"], stride=4, padding=38)
    y = F.silu(batch_norm(sd, pre + "trunk.bn1.", y, train_bn))
    for li in range(1, 5):
        for bi in range(2):
            stride = 2 if (li > 1 and bi == 0) else 1
            y = basic_block_1d(sd, f"{pre}trunk.layer{li}.{bi}.", y, stride, train_bn)
    y = F.avg_pool1d(y, 20, 20)
    return y.transpose(1, 2)


# --------------------------------------------------------------------------- losses
def ctc_loss(sd, pre, hs, hlens, ys_pad):
    """ctc.py:40-65,32-38: Linear -> (T,B,V) log_softmax -> CTCLoss(sum, zero_infinity) / B."""
    ys = [y[y != -1] for y in ys_pad]
    logits = linear(sd, pre + "ctc_lo.", hs).transpose(0, 1)
    olens = torch.tensor([len(y) for y in ys], dtype=torch.long)
    loss = F.ctc_loss(logits.log_softmax(2), torch.cat(ys), hlens.long(), olens, blank=0, reduction="sum",
                      zero_infinity=True)
    return loss / logits.shape[1]


def add_sos_eos(ys_pad, sos, eos, ignore_id=-1):
    """transformer/add_sos_eos.py:12-31."""
    ys = [y[y != ignore_id] for y in ys_pad]
    L = max(len(y) for y in ys) + 1
    ys_in = ys_pad.new_full((len(ys), L), eos)
    ys_out = ys_pad.new_full((len(ys), L), ignore_id)
    for i, y in enumerate(ys):
        ys_in[i, 0] = sos
        ys_in[i, 1 : len(y) + 1] = y
        ys_out[i, : len(y)] = y
        ys_out[i, len(y)] = eos
    return ys_in, ys_out


def label_smoothing_loss(logits, target, smoothing=0.1, ignore_id=-1):
    """label_smoothing_loss.py:41-63 with normalize_length=False: sum of KL rows / batch size."""
    B, V = logits.shape[0], logits.shape[-1]
    x = logits.reshape(-1, V)
    t = target.reshape(-1)
    ign = t == ignore_id
    td = torch.full_like(x, smoothing / (V - 1))
    td.scatter_(1, t.masked_fill(ign, 0).unsqueeze(1), 1.0 - smoothing)
    kl = F.kl_div(torch.log_softmax(x, dim=1), td, reduction="none")
    return kl.masked_fill(ign.unsqueeze(1), 0).sum() / B


def token_accuracy(logits, target, ignore_id=-1):
    """nets_utils.py:272-292."""
    pred = logits.argmax(-1)
    m = target != ignore_id
    return float((pred[m] == target[m]).sum()) / float(m.sum())


# --------------------------------------------------------------------------- the hot path
def e2e_forward(sd, x, lengths, label, modality="video", heads=12, ctc_weight=0.1, train_bn=True):
    """e2e_asr_conformer.py:63-87.  Returns (loss, loss_ctc, loss_att, acc) and the intermediates used by tests."""
    odim = sd["ctc.ctc_lo.weight"].shape[0]
    sos = eos = odim - 1
    if modality == "audio":
        lengths = torch.div(lengths, 640, rounding_mode="trunc")
        feats = audio_frontend(sd, "frontend.", x, train_bn)
    else:
        feats = video_frontend(sd, "frontend.", x, train_bn)
    T = feats.shape[1]
    pad_mask = (torch.arange(T).unsqueeze(0) < lengths.unsqueeze(1)).unsqueeze(-2)  # make_non_pad_mask, (B,1,T)
    h = linear(sd, "proj_encoder.", feats)
    enc = conformer_encoder(sd, "encoder.", h, pad_mask, heads, train_bn)
    loss_ctc = ctc_loss(sd, "ctc.", enc, lengths, label)
    ys_in, ys_out = add_sos_eos(label, sos, eos)
    L = ys_in.shape[1]
    ys_mask = (ys_in != -1).unsqueeze(-2) & torch.tril(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0)
    pred = transformer_decoder(sd, "decoder.", ys_in, ys_mask, enc, pad_mask, heads)
    loss_att = label_smoothing_loss(pred, ys_out)
    loss = ctc_weight * loss_ctc + (1 - ctc_weight) * loss_att
    acc = token_accuracy(pred, ys_out)
    return (loss, loss_ctc, loss_att, acc), dict(feats=feats, enc=enc, pred=pred, ys_in=ys_in, ys_out=ys_out)


def e2e_av_forward(sd, video, audio, lengths, label, heads=12, ctc_weight=0.1, train_bn=True):
    """Audio-visual composition (no counterpart in the reference snapshot): the two pinned stacks
    (`frontend` / `proj_encoder` / `encoder` on video, `aux_*` on audio), frame-wise concatenation, the fusion MLP
    fc2(relu(fc1(.))) -- this build's choice --, then the pinned heads exactly as in e2e_forward."""
    odim = sd["ctc.ctc_lo.weight"].shape[0]
    sos = eos = odim - 1
    vf = video_frontend(sd, "frontend.", video, train_bn)
    af = audio_frontend(sd, "aux_frontend.", audio, train_bn)
    T = min(vf.shape[1], af.shape[1])
    vf, af = vf[:, :T], af[:, :T]
    pad_mask = (torch.arange(T).unsqueeze(0) < lengths.unsqueeze(1)).unsqueeze(-2)
    v = conformer_encoder(sd, "encoder.", linear(sd, "proj_encoder.", vf), pad_mask, heads, train_bn)
    a = conformer_encoder(sd, "aux_encoder.", linear(sd, "aux_proj_encoder.", af), pad_mask, heads, train_bn)
    mem = linear(sd, "fusion.fc2.", torch.relu(linear(sd, "fusion.fc1.", torch.cat([v, a], dim=-1))))
    loss_ctc = ctc_loss(sd, "ctc.", mem, lengths, label)
    ys_in, ys_out = add_sos_eos(label, sos, eos)
    L = ys_in.shape[1]
    ys_mask = (ys_in != -1).unsqueeze(-2) & torch.tril(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0)
    pred = transformer_decoder(sd, "decoder.", ys_in, ys_mask, mem, pad_mask, heads)
    loss_att = label_smoothing_loss(pred, ys_out)
    loss = ctc_weight * loss_ctc + (1 - ctc_weight) * loss_att
    return (loss, loss_ctc, loss_att, token_accuracy(pred, ys_out)), dict(mem=mem, pred=pred)

