"""BASELINE.json configs[0] -- the plumbing check: modality=audio, 2-layer Conformer d=256 (4 heads, 2048 units) built
directly, synthetic 16 kHz wav, decoded end to end through the evaluation path lightning.ModelModule wires
(front-end -> proj -> encoder with mask None -> hybrid CTC / attention beam search), against hypotheses produced by the
REFERENCE on the CPU (tests/golden/make_golden_config0.py)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from synth import synth_state_dict  # noqa: E402

GOLD = torch.load(os.path.join(HERE, "golden", "golden_config0_v1.pt"), weights_only=False)


def wav(seconds, seed):
    g = torch.Generator().manual_seed(900 + seed)
    w = torch.randn(int(16000 * seconds), generator=g)
    return ((w - w.mean()) / w.std()).unsqueeze(1)


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f"{c['seconds']:.0f}s")
def test_config0_audio_2layer_d256_decode(dev, case):
    import lightning
    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.e2e import E2E

    if dev.type == "cpu" and case["seconds"] > 2:
        pytest.skip("the 4 s clip runs on the MI355X only (emulator time)")
    c = GOLD
    AF.invalidate_weight_cache()
    m = E2E(c["odim"], "audio", adim=c["D"], aheads=c["H"], eunits=c["U"], elayers=c["nenc"], dunits=c["U"], dlayers=c["ndec"])
    m.load_state_dict(synth_state_dict(m.state_dict(), c["seed"]), strict=True)
    m.to(dev).eval()
    x = wav(case["seconds"], int(case["seconds"])).to(dev)
    bs = lightning.get_beam_search_decoder(m, [str(i) for i in range(c["odim"])], beam_size=case["beam"])
    was = AF._state["precise"]
    AF.set_precise(True)
    try:
        with torch.no_grad():
            feats = m.proj_encoder(m.frontend(x.unsqueeze(0)))
            enc, _ = m.encoder(feats, None)
            assert enc.shape[1] == case["frames"]
            ref = case["enc_sample"]
            assert (enc[0, :, :8].float().cpu() - ref).abs().max() < 1e-3 * float(ref.abs().max())
            nbest = bs(enc.squeeze(0).float())
    finally:
        AF.set_precise(was)
        AF.invalidate_weight_cache()
    assert len(nbest) == case["n_ended"]
    for got, want in zip(nbest, case["hyps"]):
        d = got.asdict()
        assert d["yseq"] == want["yseq"]
        assert abs(d["score"] - want["score"]) < 1e-3 * max(1.0, abs(want["score"]))
