"""Golden vectors for the evaluation path: runs the REFERENCE BatchBeamSearch (/root/reference, PyTorch CPU) with the
scorers lightning.get_beam_search_decoder wires (decoder + CTCPrefixScorer + LengthBonus) on small synthetic
decoder / CTC instances and a synthetic encoder output, and also the reference CTCPrefixScoreTH alone.
Run in the build container only:   python tests/golden/make_golden_decode.py   ->  tests/golden/golden_decode_v1.pt"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
from synth import synth_state_dict  # noqa: E402

from espnet.nets.batch_beam_search import BatchBeamSearch  # noqa: E402
from espnet.nets.ctc_prefix_score import CTCPrefixScoreTH  # noqa: E402
from espnet.nets.pytorch_backend.ctc import CTC  # noqa: E402
from espnet.nets.pytorch_backend.decoder.transformer_decoder import TransformerDecoder  # noqa: E402
from espnet.nets.scorers.ctc import CTCPrefixScorer  # noqa: E402
from espnet.nets.scorers.length_bonus import LengthBonus  # noqa: E402


def beam_case(seed, odim, T, beam, ctc_weight, penalty, D=128):
    torch.manual_seed(0)
    dec = TransformerDecoder(odim, attention_dim=D, attention_heads=2, linear_units=256, num_blocks=2).eval()
    ctc = CTC(odim, D, 0.1, reduce=True).eval()
    dec.load_state_dict(synth_state_dict(dec.state_dict(), seed))
    ctc.load_state_dict(synth_state_dict(ctc.state_dict(), seed + 1))
    g = torch.Generator().manual_seed(500 + seed)
    enc = torch.randn(T, D, generator=g) * 1.5
    token_list = [str(i) for i in range(odim)]
    scorers = {"decoder": dec, "ctc": CTCPrefixScorer(ctc=ctc, eos=odim - 1), "lm": None,
               "length_bonus": LengthBonus(len(token_list))}
    weights = {"decoder": 1.0 - ctc_weight, "ctc": ctc_weight, "lm": 0.0, "length_bonus": penalty}
    bs = BatchBeamSearch(beam_size=beam, vocab_size=odim, weights=weights, scorers=scorers, sos=odim - 1, eos=odim - 1,
                         token_list=token_list, pre_beam_score_key=None if ctc_weight == 1.0 else "decoder")
    with torch.no_grad():
        nbest = bs(enc)
    hyps = [h.asdict() for h in nbest[:5]]
    return dict(seed=seed, odim=odim, T=T, beam=beam, ctc_weight=ctc_weight, penalty=penalty, D=D, n_ended=len(nbest),
                hyps=[dict(yseq=h["yseq"], score=h["score"], scores=h["scores"]) for h in hyps])


def prefix_case(seed, T, V, NH, S):
    """CTCPrefixScoreTH for two consecutive steps of NH hypotheses with S candidates each."""
    g = torch.Generator().manual_seed(900 + seed)
    logp = torch.log_softmax(torch.randn(1, T, V, generator=g) * 2.0, -1)
    impl = CTCPrefixScoreTH(logp.clone(), torch.tensor([T]), 0, V - 1)
    y0 = [[V - 1]]
    ids0 = torch.stack([torch.randperm(V, generator=g)[:S]])
    sc0, st0 = impl(y0, None, ids0)
    # extend the single hypothesis by NH different tokens taken from its candidates
    toks = ids0[0, :NH]
    r, log_psi, fmin, fmax, idmap = st0
    state1 = (torch.stack([r[:, :, 0, idmap[0, t]] for t in toks], dim=2),
              torch.stack([log_psi[0, t].expand(V) for t in toks]), fmin, fmax)
    y1 = [[V - 1, int(t)] for t in toks]
    ids1 = torch.stack([torch.randperm(V, generator=g)[:S] for _ in range(NH)])
    for n in range(NH):  # make sure the "same as last token" branch is exercised (candidates stay unique, as from top-k)
        hit = (ids1[n] == toks[n]).nonzero()
        if len(hit):
            ids1[n, hit[0, 0]] = ids1[n, 0]
        ids1[n, 0] = toks[n]
    sc1, st1 = impl(y1, state1, ids1)
    return dict(seed=seed, T=T, V=V, NH=NH, S=S, logp=logp[0].clone(), ids0=ids0, sc0=sc0, toks=toks, ids1=ids1, sc1=sc1,
                r1=st1[0].clone())


def e2e_decode_case(seed, T, beam):
    """Full-size video E2E in eval mode: front-end -> encoder -> beam search exactly as lightning.ModelModule.forward
    (lightning.py:54-64, 126-158) does it."""
    from synth import synth_batch

    from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E

    torch.manual_seed(0)
    m = E2E(5049, "video").eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed))
    x, _, _ = synth_batch("video", 1, T, 3, 5049, seed=seed, lengths=[T])
    token_list = [str(i) for i in range(5049)]
    scorers = m.scorers()
    scorers["lm"] = None
    scorers["length_bonus"] = LengthBonus(len(token_list))
    weights = {"decoder": 0.9, "ctc": 0.1, "lm": 0.0, "length_bonus": 0}
    bs = BatchBeamSearch(beam_size=beam, vocab_size=5049, weights=weights, scorers=scorers, sos=5048, eos=5048,
                         token_list=token_list, pre_beam_score_key="decoder")
    with torch.no_grad():
        feats = m.proj_encoder(m.frontend(x))
        enc, _ = m.encoder(feats, None)
        nbest = bs(enc.squeeze(0))
    hyps = [h.asdict() for h in nbest[:3]]
    return dict(seed=seed, T=T, beam=beam, n_ended=len(nbest), enc_sample=enc[0, :, :8].clone(),
                hyps=[dict(yseq=h["yseq"], score=h["score"]) for h in hyps])


if __name__ == "__main__":
    out = {"beam": [beam_case(1, 40, 15, 5, 0.1, 0.0), beam_case(2, 50, 23, 8, 0.3, 0.5), beam_case(3, 30, 9, 4, 0.1, 0.0),
                    beam_case(4, 64, 31, 10, 0.1, 0.0)],
           "prefix": [prefix_case(1, 12, 20, 3, 6), prefix_case(2, 25, 33, 5, 9)],
           "e2e": [e2e_decode_case(5, 14, 10)]}
    torch.save(out, os.path.join(HERE, "golden_decode_v1.pt"))
    print("e2e", out["e2e"][0]["n_ended"], out["e2e"][0]["hyps"][0])
    for c in out["beam"]:
        print(c["seed"], c["n_ended"], [(h["yseq"], round(h["score"], 4)) for h in c["hyps"][:2]])
