"""Golden for BASELINE.json configs[0] (plumbing check): modality=audio, 2-layer Conformer d=256 / 4 heads / 2048 units built
directly (the reference's E2E hard-codes the 768/12 sizes, SURVEY F3: its sub-modules are replaced the way F3 describes),
synthetic 16 kHz wav, eval mode, the reference's own decode path (lightning.py:54-64,126-158): front-end -> proj -> encoder
(mask None) -> BatchBeamSearch(decoder 0.9 + CTC prefix 0.1 + length bonus) on the CPU.
Run in the build container only:   python tests/golden/make_golden_config0.py   ->  tests/golden/golden_config0_v1.pt"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
from synth import synth_state_dict  # noqa: E402

from espnet.nets.batch_beam_search import BatchBeamSearch  # noqa: E402
from espnet.nets.pytorch_backend.ctc import CTC  # noqa: E402
from espnet.nets.pytorch_backend.decoder.transformer_decoder import TransformerDecoder  # noqa: E402
from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E  # noqa: E402
from espnet.nets.pytorch_backend.encoder.conformer_encoder import ConformerEncoder  # noqa: E402
from espnet.nets.scorers.length_bonus import LengthBonus  # noqa: E402

ODIM, D, H, U, NENC, NDEC, SEED = 61, 256, 4, 2048, 2, 2, 23


def build():
    torch.manual_seed(0)
    m = E2E(ODIM, "audio")
    m.proj_encoder = torch.nn.Linear(512, D)
    m.encoder = ConformerEncoder(attention_dim=D, attention_heads=H, linear_units=U, num_blocks=NENC)
    m.decoder = TransformerDecoder(odim=ODIM, attention_dim=D, attention_heads=H, linear_units=U, num_blocks=NDEC)
    m.ctc = CTC(ODIM, D, 0.1, reduce=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), SEED))
    return m.eval()


def wav(seconds, seed):
    g = torch.Generator().manual_seed(900 + seed)
    w = torch.randn(int(16000 * seconds), generator=g)
    return ((w - w.mean()) / w.std()).unsqueeze(1)  # (T, 1), layer-normed like AudioTransform's last stage


if __name__ == "__main__":
    m = build()
    cases = []
    for seconds, beam in ((2.0, 5), (4.0, 8)):
        x = wav(seconds, int(seconds))
        token_list = [str(i) for i in range(ODIM)]
        scorers = m.scorers()
        scorers["lm"] = None
        scorers["length_bonus"] = LengthBonus(len(token_list))
        bs = BatchBeamSearch(beam_size=beam, vocab_size=ODIM, weights={"decoder": 0.9, "ctc": 0.1, "lm": 0.0, "length_bonus": 0.0},
                             scorers=scorers, sos=ODIM - 1, eos=ODIM - 1, token_list=token_list, pre_beam_score_key="decoder")
        with torch.no_grad():
            feats = m.proj_encoder(m.frontend(x.unsqueeze(0)))
            enc, _ = m.encoder(feats, None)
            nbest = bs(enc.squeeze(0))
        hyps = [h.asdict() for h in nbest[:3]]
        cases.append(dict(seconds=seconds, beam=beam, frames=enc.shape[1], enc_sample=enc[0, :, :8].clone(), n_ended=len(nbest),
                          hyps=[dict(yseq=h["yseq"], score=h["score"]) for h in hyps]))
        print(seconds, enc.shape, hyps[0]["yseq"][:12], hyps[0]["score"])
    torch.save(dict(odim=ODIM, D=D, H=H, U=U, nenc=NENC, ndec=NDEC, seed=SEED, cases=cases), os.path.join(HERE, "golden_config0_v1.pt"))
