"""Audio-visual fusion model (BASELINE config 5).

The reference snapshot contains NO audio-visual model (`--modality` is audio or video; its README defers the AV
code to a later release), so there is nothing to pin against: this is the composition the survey names -- two front-end +
encoder stacks, a fusion MLP, the shared CTC head / Transformer decoder / joint loss of E2E -- built from the same modules
and kernels as the single-modality hot path.  Parity is claimed only for the parts that exist in the reference (each
stack, the heads: the oracle functions they are checked against are pinned to the reference); the fusion head
(Linear(2D, hidden) -> ReLU -> Linear(hidden, D) on the frame-wise concatenation) is this build's choice.  Parameter names:
the video stack keeps E2E's names (`frontend`, `proj_encoder`, `encoder`), the audio stack is `aux_*`, the head `fusion`.
"""
import torch
from torch import nn

from . import functional as AF
from . import nets
from . import ops
from .frontend import audio_resnet, video_resnet


class FusionMLP(nn.Module):
    def __init__(self, idim, hdim, odim):
        super().__init__()
        self.fc1 = nn.Linear(idim, hdim)
        self.fc2 = nn.Linear(hdim, odim)

    def forward(self, x):
        return AF.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)


class E2EAV(nn.Module):
    def __init__(self, odim, ctc_weight=0.1, ignore_id=-1, *, adim=768, aheads=12, eunits=3072, elayers=12, dunits=3072,
                 dlayers=6, cnn_module_kernel=31, fusion_hdim=8192):
        super().__init__()
        enc = dict(attention_dim=adim, attention_heads=aheads, linear_units=eunits, num_blocks=elayers,
                   cnn_module_kernel=cnn_module_kernel)
        self.frontend = video_resnet()
        self.proj_encoder = nn.Linear(512, adim)
        self.encoder = nets.ConformerEncoder(**enc)
        self.aux_frontend = audio_resnet()
        self.aux_proj_encoder = nn.Linear(512, adim)
        self.aux_encoder = nets.ConformerEncoder(**enc)
        self.fusion = FusionMLP(2 * adim, fusion_hdim, adim)
        self.decoder = nets.TransformerDecoder(odim=odim, attention_dim=adim, attention_heads=aheads,
                                               linear_units=dunits, num_blocks=dlayers)
        self.blank = 0
        self.sos = self.eos = odim - 1
        self.odim = odim
        self.ignore_id = ignore_id
        self.ctc_weight = ctc_weight
        self.ctc = nets.CTC(odim, adim, 0.1, reduce=True)
        self.criterion = nets.LabelSmoothingLoss(odim, ignore_id, 0.1, False)

    def encode(self, video, audio, lengths):
        """video (B, T, 1, 88, 88), audio (B, 640 T, 1), lengths in video frames -> fused memory (B, T, D), mask."""
        vf = self.frontend(video)
        af = self.aux_frontend(audio)
        T = min(vf.shape[1], af.shape[1])  # 640 samples per frame: equal up to the last partial frame
        vf, af = vf[:, :T], af[:, :T]
        lengths = lengths.to(vf.device)
        mask = nets.non_pad_mask_device(lengths, T)
        v, _ = self.encoder(AF.linear(vf, self.proj_encoder.weight, self.proj_encoder.bias, out_dtype=torch.float32), mask)
        a, _ = self.aux_encoder(AF.linear(af, self.aux_proj_encoder.weight, self.aux_proj_encoder.bias,
                                          out_dtype=torch.float32), mask)
        return self.fusion(torch.cat([v, a], dim=-1)), mask  # the concatenation is data movement only

    def forward_tensors(self, video, audio, lengths, label):
        mem, mask = self.encode(video, audio, lengths)
        lengths = lengths.to(mem.device)
        loss_ctc, _ = self.ctc(mem, lengths, label)
        ys_in, ys_out, ys_mask, n_tok = ops.prepare_targets(label.to(mem.device), self.sos, self.eos, self.ignore_id)
        pred, _ = self.decoder(ys_in, ys_mask, mem, mask)
        loss_att = self.criterion(pred, ys_out)
        loss = self.ctc_weight * loss_ctc + (1 - self.ctc_weight) * loss_att
        return loss, loss_ctc, loss_att, self.criterion.last_hits, n_tok[0]

    def forward(self, video, audio, lengths, label):
        loss, loss_ctc, loss_att, hits, n_tok = self.forward_tensors(video, audio, lengths, label)
        return loss, loss_ctc, loss_att, float(hits.detach()) / float(n_tok)
