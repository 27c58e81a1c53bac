"""Evaluation path: CTC prefix-score kernel and the batched hybrid CTC/attention beam
search against golden vectors produced by the reference's CTCPrefixScoreTH / BatchBeamSearch
(tests/golden/make_golden_decode.py)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from synth import synth_state_dict  # noqa: E402

from auto_avsr_amd import functional as AF  # noqa: E402
from auto_avsr_amd import nets  # noqa: E402
from auto_avsr_amd.decoding import LOGZERO, BatchBeamSearch, CTCPrefixScorer, LengthBonus, end_detect  # noqa: E402

GOLD = torch.load(os.path.join(HERE, "golden", "golden_decode_v1.pt"), weights_only=False)


class _FixedCtc(torch.nn.Module):
    """CTC head stand-in that returns a given log-softmax (the kernel is tested on the golden posteriors themselves)."""

    def __init__(self, logp):
        super().__init__()
        self.logp = logp
        self.ctc_lo = torch.nn.Linear(1, logp.shape[-1])

    def log_softmax(self, x):
        return self.logp.unsqueeze(0)


@pytest.mark.parametrize("case", GOLD["prefix"], ids=lambda c: f"seed{c['seed']}")
def test_ctc_prefix_scores_vs_reference(dev, case):
    T, V, NH, S = case["T"], case["V"], case["NH"], case["S"]
    sc = CTCPrefixScorer(_FixedCtc(case["logp"].to(dev)), eos=V - 1)
    st = sc.batch_init_state(torch.zeros(T, 1, device=dev))
    y0 = torch.tensor([[V - 1]], device=dev)
    s0, full0 = sc.batch_score_partial(y0, case["ids0"].to(dev), st)
    ref0 = case["sc0"]
    assert (s0.cpu() - ref0).abs().max() < 1e-3 * max(1.0, float(ref0[ref0 > LOGZERO / 2].abs().max()))
    toks = case["toks"].to(dev)
    st1 = sc.select_states(full0, torch.zeros(NH, dtype=torch.int64, device=dev), toks)
    y1 = torch.cat([torch.full((NH, 1), V - 1, device=dev), toks.unsqueeze(1)], 1)
    s1, full1 = sc.batch_score_partial(y1, case["ids1"].to(dev), st1)
    ref1 = case["sc1"]
    assert (s1.cpu() - ref1).abs().max() < 1e-3 * max(1.0, float(ref1[ref1 > LOGZERO / 2].abs().max()))
    r1 = full1[0].cpu()
    live = case["r1"] > LOGZERO / 2
    assert ((r1 - case["r1"]).abs()[live]).max() < 1e-3 * float(case["r1"][live].abs().max())
    assert bool((r1[~live] < LOGZERO / 2).all())


def _search(dev, case, native, maxlenratio=0.0, beam=None, linear_units=256):
    from auto_avsr_amd import decoding

    odim, T, D = case["odim"], case["T"], case["D"]
    torch.manual_seed(0)
    dec = nets.TransformerDecoder(odim, attention_dim=D, attention_heads=2, linear_units=linear_units, num_blocks=2).eval()
    ctc = nets.CTC(odim, D, 0.1, reduce=True).eval()
    dec.load_state_dict(synth_state_dict(dec.state_dict(), case["seed"]))
    ctc.load_state_dict(synth_state_dict(ctc.state_dict(), case["seed"] + 1))
    dec, ctc = dec.to(dev), ctc.to(dev)
    g = torch.Generator().manual_seed(500 + case["seed"])
    enc = (torch.randn(T, D, generator=g) * 1.5).to(dev)
    scorers = {"decoder": dec, "ctc": CTCPrefixScorer(ctc, odim - 1), "lm": None, "length_bonus": LengthBonus(odim)}
    weights = {"decoder": 1.0 - case["ctc_weight"], "ctc": case["ctc_weight"], "lm": 0.0, "length_bonus": case["penalty"]}
    bs = BatchBeamSearch(beam_size=beam or case["beam"], vocab_size=odim, weights=weights, scorers=scorers, sos=odim - 1,
                         eos=odim - 1, token_list=[str(i) for i in range(odim)], pre_beam_score_key="decoder")
    was = decoding.NATIVE_BEAM
    decoding.NATIVE_BEAM = native
    AF.set_precise(True)
    try:
        nbest = bs(enc, maxlenratio=maxlenratio)
    finally:
        AF.set_precise(False)
        decoding.NATIVE_BEAM = was
    assert bool(bs._native) == native  # the path asked for is the one that ran
    return nbest


@pytest.mark.parametrize("native", [True, False], ids=["native", "python"])
@pytest.mark.parametrize("case", GOLD["beam"], ids=lambda c: f"seed{c['seed']}")
def test_beam_search_vs_reference(dev, case, native):
    """Same weights, same encoder output: the n-best token sequences equal the reference's, scores within 1e-3 -- for the
    one-call-per-step search (csrc/decode.hip through decode_native.py, the default) and for the python-issued step."""
    nbest = _search(dev, case, native)
    assert len(nbest) == case["n_ended"]
    for got, ref in zip(nbest, case["hyps"]):
        d = got.asdict()
        assert d["yseq"] == ref["yseq"]
        assert abs(d["score"] - ref["score"]) < 1e-3 * max(1.0, abs(ref["score"]))
        for k, v in ref["scores"].items():
            assert abs(d["scores"][k] - v) < 2e-3 * max(1.0, abs(v)), k


@pytest.mark.parametrize("maxlenratio", [0.0, -4, 0.5])
def test_native_beam_search_equals_python_step(dev, maxlenratio):
    """Every ended hypothesis (not only the reference's recorded n-best), every per-scorer score, with a search that runs
    into the forced end (maxlenratio = -4: four steps, beam_search.py:430-436) and one stopped by the end-detection rule.
    Hypotheses the CTC scorer rules out (LOGZERO = -1e10 in the score: more labels than frames) tie at ~-1e9 and are ordered
    arbitrarily by any top-k -- excluded from the comparison."""
    case = GOLD["beam"][3]
    a, b = _search(dev, case, True, maxlenratio), _search(dev, case, False, maxlenratio)
    assert len(a) == len(b) and len(a) >= 1
    live = 0
    for x, y in zip(a, b):
        x, y = x.asdict(), y.asdict()
        if y["score"] < -1e8:
            assert x["score"] < -1e8
            continue
        live += 1
        assert x["yseq"] == y["yseq"]
        assert abs(x["score"] - y["score"]) < 1e-3 * max(1.0, abs(y["score"]))
        assert set(x["scores"]) == set(y["scores"])
        for k, v in y["scores"].items():
            assert abs(x["scores"][k] - v) < 1e-3 * max(1.0, abs(v)), k
    assert live >= 3


@pytest.mark.parametrize("M,N,K,ln,act,res", [(5, 64, 128, False, 0, False), (5, 64, 128, True, 0, False), (40, 96, 128, True, 1, True),
                                              (40, 50, 256, False, 0, True), (33, 128, 768, True, 0, True), (7, 128, 2048, False, 0, True),
                                              (50, 64, 128, True, 0, True), (1, 5049, 128, True, 0, False)])
def test_decode_linear(dev, M, N, K, ln, act, res):
    """The linear layer of a decoding step (csrc/decode.hip skinny16_kernel: MFMA fragments straight from global memory, split-plane
    MFMAs, LayerNorm from the producer's row statistics, K slices + row-sum for long contractions) against float64, and the row
    statistics it leaves for the next LayerNorm."""
    import ctypes

    from auto_avsr_amd import _lib, ops

    L = _lib.lib()
    g0 = torch.Generator().manual_seed(M * 1000 + N + K)
    rnd = lambda *sh: torch.randn(*sh, generator=g0)  # noqa: E731
    A, W, b = (rnd(M, K) * 2 + 0.5).to(dev), (rnd(N, K) / K ** 0.5).to(dev), rnd(N).to(dev)
    g, be, R = (torch.rand(K, generator=g0) + 0.5).to(dev), (rnd(K) * 0.1).to(dev), rnd(M, N).to(dev)
    C = torch.zeros(M, N, device=dev)
    nt_max = (N + 15) // 16
    st = torch.zeros(M * nt_max * 2, device=dev)
    part = torch.zeros(8 * M * N, device=dev)
    st_in = torch.stack([A.sum(1), (A * A).sum(1)], 1).contiguous()
    n = ctypes.c_int(0)
    L.call("avsr_decode_linear", A.data_ptr(), K, W.data_ptr(), M, N, K, b.data_ptr(), g.data_ptr() if ln else None,
           be.data_ptr() if ln else None, 1e-12, st_in.data_ptr() if ln else None, 1, act, R.data_ptr() if res else None, N,
           C.data_ptr(), N, st.data_ptr(), ctypes.cast(ctypes.pointer(n), ctypes.c_void_p), part.data_ptr(), ops._stream(A))
    X = torch.nn.functional.layer_norm(A.double(), (K,), g.double(), be.double(), 1e-12) if ln else A.double()
    ref = X @ W.double().T + b.double()
    if act:
        ref = ref.relu()
    if res:
        ref = ref + R.double()
    assert float((C.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert n.value == (1 if K > 768 else nt_max)
    stv = st[: M * n.value * 2].view(M, n.value, 2).double()
    assert float((stv[:, :, 0].sum(1) - ref.sum(1)).abs().max()) < 1e-3 * float(ref.abs().sum(1).max())
    assert float((stv[:, :, 1].sum(1) - (ref * ref).sum(1)).abs().max()) < 1e-4 * float((ref * ref).sum(1).max())


@pytest.mark.parametrize("T", [80, 300])
def test_native_beam_search_long_memory(dev, T):
    """Encoder memories longer than one attention chunk: the source-attention blocks run with 8 (T <= 256) and 16 waves."""
    case = dict(GOLD["beam"][0], T=T)
    a, b = _search(dev, case, True, maxlenratio=-5), _search(dev, case, False, maxlenratio=-5)
    assert len(a) == len(b) and len(a) >= 1
    for x, y in zip(a, b):
        x, y = x.asdict(), y.asdict()
        if y["score"] < -1e8:
            continue
        assert x["yseq"] == y["yseq"]
        assert abs(x["score"] - y["score"]) < 1e-3 * max(1.0, abs(y["score"]))


def test_native_beam_search_long_ffn(dev):
    """linear_units = 2048 (the reference model's decoder width): the FFN's second contraction runs as K slices across blocks +
    the row-sum kernel (csrc/decode.hip skinny()), which the 256-unit golden cases never reach."""
    case = GOLD["beam"][0]
    a, b = _search(dev, case, True, linear_units=2048), _search(dev, case, False, linear_units=2048)
    assert len(a) == len(b) and len(a) >= 1
    for x, y in zip(a, b):
        x, y = x.asdict(), y.asdict()
        if y["score"] < -1e8:
            continue
        assert x["yseq"] == y["yseq"]
        assert abs(x["score"] - y["score"]) < 1e-3 * max(1.0, abs(y["score"]))


@pytest.mark.parametrize("beam,T,maxlenratio", [(2, 9, 0.0), (5, 1, 0.0), (5, 9, -1), (4, 2, 0.0)])
def test_native_beam_search_edge_cases(dev, beam, T, maxlenratio):
    """The smallest beam the library takes (2: pre-beam 3), a one-frame utterance, a one-step search (forced end at the first
    step), a two-frame utterance: same ended hypotheses as the python-issued step."""
    case = dict(GOLD["beam"][2], T=T)
    a, b = _search(dev, case, True, maxlenratio, beam=beam), _search(dev, case, False, maxlenratio, beam=beam)
    assert len(a) == len(b) and len(a) >= 1
    for x, y in zip(a, b):
        x, y = x.asdict(), y.asdict()
        if y["score"] < -1e8:
            assert x["score"] < -1e8
            continue
        assert x["yseq"] == y["yseq"]
        assert abs(x["score"] - y["score"]) < 1e-3 * max(1.0, abs(y["score"]))


def test_forward_many_equals_one_at_a_time(dev):
    """BatchBeamSearch.forward_many: five utterances of different lengths through three concurrent sessions (host threads + streams)
    give the hypotheses of five separate calls."""
    case = GOLD["beam"][1]
    odim, D = case["odim"], case["D"]
    torch.manual_seed(0)
    dec = nets.TransformerDecoder(odim, attention_dim=D, attention_heads=2, linear_units=256, num_blocks=2).eval()
    ctc = nets.CTC(odim, D, 0.1, reduce=True).eval()
    dec.load_state_dict(synth_state_dict(dec.state_dict(), case["seed"]))
    ctc.load_state_dict(synth_state_dict(ctc.state_dict(), case["seed"] + 1))
    dec, ctc = dec.to(dev), ctc.to(dev)
    g = torch.Generator().manual_seed(77)
    xs = [(torch.randn(T, D, generator=g) * 1.5).to(dev) for T in (9, 17, 12, 23, 15)]
    scorers = {"decoder": dec, "ctc": CTCPrefixScorer(ctc, odim - 1), "lm": None, "length_bonus": LengthBonus(odim)}
    weights = {"decoder": 0.7, "ctc": 0.3, "lm": 0.0, "length_bonus": 0.5}
    bs = BatchBeamSearch(beam_size=6, vocab_size=odim, weights=weights, scorers=scorers, sos=odim - 1, eos=odim - 1,
                         token_list=None, pre_beam_score_key="decoder")
    AF.set_precise(True)
    try:
        many = bs.forward_many(xs, workers=3)
        assert bs._native and len(bs._native_pool) == 3
        single = [bs(x) for x in xs]
    finally:
        AF.set_precise(False)
    assert len(many) == len(single) == 5
    for a, b in zip(many, single):
        assert len(a) == len(b) and len(a) >= 1
        for x, y in zip(a, b):
            x, y = x.asdict(), y.asdict()
            if y["score"] < -1e8:
                continue
            assert x["yseq"] == y["yseq"] and abs(x["score"] - y["score"]) < 1e-4 * max(1.0, abs(y["score"]))


def test_native_beam_refuses_what_it_cannot_score(dev):
    """Scorer sets outside the reference's wiring stay on the python step: a vocabulary smaller than the pre-beam (no pre-beam,
    beam_search.py:85-90) and a foreign full scorer."""
    from auto_avsr_amd.decode_native import NativeBeam

    case = dict(GOLD["beam"][0])
    nb = _search(dev, case, False, beam=30)  # pre-beam 45 >= vocabulary 40: do_pre_beam is False
    assert len(nb) >= 1
    dec = nets.TransformerDecoder(40, attention_dim=128, attention_heads=2, linear_units=256, num_blocks=1).eval()
    ctc = nets.CTC(40, 128, 0.1, reduce=True).eval()

    class Other(LengthBonus):
        pass

    mk = lambda sc, beam: BatchBeamSearch(beam_size=beam, vocab_size=40, weights={k: 0.5 for k in sc}, scorers=sc, sos=39, eos=39,  # noqa: E731
                                          token_list=None, pre_beam_score_key="decoder")
    assert NativeBeam.supported(mk({"decoder": dec, "ctc": CTCPrefixScorer(ctc, 39), "length_bonus": LengthBonus(40)}, 5))
    assert not NativeBeam.supported(mk({"decoder": dec, "ctc": CTCPrefixScorer(ctc, 39), "lm": Other(40)}, 5))
    assert not NativeBeam.supported(mk({"decoder": dec, "ctc": CTCPrefixScorer(ctc, 39)}, 30))
    assert not NativeBeam.supported(mk({"decoder": dec, "length_bonus": LengthBonus(40)}, 5))


def test_end_detect_rule():
    ended = [dict(yseq=[0] * 5, score=-1.0), dict(yseq=[0] * 6, score=-20.0), dict(yseq=[0] * 7, score=-20.0),
             dict(yseq=[0] * 8, score=-20.0)]
    assert end_detect(ended, 8) and not end_detect(ended, 7) and not end_detect([], 3)


def test_e2e_scorers_and_decoder_factory(dev):
    """E2E.scorers() + lightning.get_beam_search_decoder produce a working search on a small instance."""
    sys.path.insert(0, os.path.dirname(HERE))
    import lightning
    from auto_avsr_amd.e2e import E2E

    m = E2E(40, "video", adim=128, aheads=2, eunits=256, elayers=1, dunits=256, dlayers=1, cnn_module_kernel=7).to(dev).eval()
    sc = m.scorers()
    assert set(sc) == {"decoder", "ctc"}
    bs = lightning.get_beam_search_decoder(m, [str(i) for i in range(40)], beam_size=3)
    x = torch.randn(6, 1, 88, 88, device=dev)
    with torch.no_grad():
        feats = m.proj_encoder(m.frontend(x.unsqueeze(0)))
        enc, _ = m.encoder(feats, None)
        nbest = bs(enc.squeeze(0))
    assert len(nbest) >= 1 and nbest[0].asdict()["yseq"][0] == 39 and nbest[0].asdict()["yseq"][-1] == 39
