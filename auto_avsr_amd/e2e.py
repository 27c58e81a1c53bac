"""E2E: front-end -> Linear(512, D) -> Conformer encoder -> {CTC, Transformer decoder + label smoothing}.

Mirror of espnet/nets/pytorch_backend/e2e_asr_conformer.py:21-87.  The constructor keeps the reference
signature ``E2E(odim, modality, ctc_weight=0.1, ignore_id=-1)`` and its hard-coded 768/12/3072/12+6 sizes as
defaults; the extra keyword arguments (not in the reference) only exist so that tests can build
small instances of the same graph."""
import torch
from torch import nn

from . import functional as AF
from . import nets
from . import ops
from .frontend import audio_resnet, video_resnet


class E2E(nn.Module):
    def __init__(self, odim, modality, ctc_weight=0.1, ignore_id=-1, *, adim=768, aheads=12, eunits=3072, elayers=12,
                 dunits=3072, dlayers=6, cnn_module_kernel=31):
        super().__init__()
        self.modality = modality
        if modality == "audio":
            self.frontend = audio_resnet()
        elif modality == "video":
            self.frontend = video_resnet()
        self.proj_encoder = nn.Linear(512, adim)
        self.encoder = nets.ConformerEncoder(attention_dim=adim, attention_heads=aheads, linear_units=eunits,
                                             num_blocks=elayers, cnn_module_kernel=cnn_module_kernel)
        self.decoder = nets.TransformerDecoder(odim=odim, attention_dim=adim, attention_heads=aheads,
                                               linear_units=dunits, num_blocks=dlayers)
        self.blank = 0
        self.sos = odim - 1
        self.eos = odim - 1
        self.odim = odim
        self.ignore_id = ignore_id
        self.ctc_weight = ctc_weight
        self.ctc = nets.CTC(odim, adim, 0.1, reduce=True)
        self.criterion = nets.LabelSmoothingLoss(self.odim, self.ignore_id, 0.1, False)

    def _side_stream(self, t):
        """The second stream of the training step on t's device (None on the CPU / emulator)."""
        if t.device.type != "cuda":
            return None
        st = getattr(self, "_side_streams", None)
        if st is None:
            st = self._side_streams = {}
        if t.device not in st:
            st[t.device] = torch.cuda.Stream(t.device)
        return st[t.device]

    def scorers(self):
        """e2e_asr_conformer.py:60-61: the scorers of hybrid CTC / attention beam search."""
        from .decoding import CTCPrefixScorer

        return dict(decoder=self.decoder, ctc=CTCPrefixScorer(self.ctc, self.eos))

    def forward_tensors(self, x, lengths, label):
        """The training hot path with no host synchronisation: returns device scalars
        (loss, loss_ctc, loss_att, n_correct, n_tokens)."""
        if self.modality == "audio":
            lengths = torch.div(lengths, 640, rounding_mode="trunc")
        feats = self.frontend(x)
        lengths = lengths.to(feats.device)
        padding_mask = nets.non_pad_mask_device(lengths, feats.shape[1])
        h = AF.linear(feats, self.proj_encoder.weight, self.proj_encoder.bias, out_dtype=torch.float32)
        enc, _ = self.encoder(h, padding_mask)
        # The CTC branch and the decoder branch share nothing between here and the weighted sum (e2e_asr_conformer.py:130-147): the
        # CTC branch runs on a second stream (functional._SIDE_BRANCH), forward and -- autograd keeps a node on its forward's
        # stream -- backward, beside the decoder's many small launches
        side = self._side_stream(enc) if (AF._SIDE_BRANCH and torch.is_grad_enabled()) else None
        if side is not None:
            main = torch.cuda.current_stream(enc.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                loss_ctc, _ = self.ctc(enc, lengths, label)
        else:
            loss_ctc, _ = self.ctc(enc, lengths, label)
        # add_sos_eos + target_mask (e2e_asr_conformer.py:138-139) with the static width Lmax + 1, and the token count of
        # th_accuracy's denominator: one launch (csrc/loss.hip prepare_targets_kernel; nets.add_sos_eos_static /
        # nets.target_mask are the torch statement of the same, kept as the test reference)
        ys_in, ys_out, ys_mask, n_tok = ops.prepare_targets(label.to(feats.device), self.sos, self.eos, self.ignore_id)
        pred, _ = self.decoder(ys_in, ys_mask, enc, padding_mask)
        loss_att = self.criterion(pred, ys_out)
        if side is not None:
            main.wait_stream(side)
        loss = self.ctc_weight * loss_ctc + (1 - self.ctc_weight) * loss_att
        return loss, loss_ctc, loss_att, self.criterion.last_hits, n_tok[0]

    def forward(self, x, lengths, label):
        loss, loss_ctc, loss_att, hits, n_tok = self.forward_tensors(x, lengths, label)
        acc = float(hits.detach()) / float(n_tok)  # the reference returns a python float too (nets_utils.py:292)
        return loss, loss_ctc, loss_att, acc
