"""Parity at the BENCHMARKED shapes: full-size video model on the survey's batch A (4 x 400 frames,
64 labels) and batch B (16 x 100, 16 labels) against numbers produced by the REFERENCE implementation
(tests/golden/make_golden_bench.py -> golden_bench_v1.pt), in ALL FOUR numerical modes:

* precise (split-bf16 contractions forward and backward) and hpf (the same forward, bf16 backward): logits / CTC log-probs at
  1e-5, far inside the north-star bound of 1e-3 relative;
* mixed (the mode bench.py times by default: f16 / split-plane forward per component, bf16 backward): logits / CTC log-probs
  asserted inside 1e-3;
* bf16 (bench.py's secondary `bf16` object): measured and printed; bounds state what 8-bit significands deliver at this depth,
  and the same numbers appear in bench.py's `parity` blocks.

Gradients are checked element-wise on 64 sampled entries per tensor (+ cosine, + norm), not by norm alone."""
import json
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import bench_common as BC  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return torch.load(BC.FIXTURE, weights_only=False)


def _model(seed, modality="video"):
    from auto_avsr_amd import _lib
    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.e2e import E2E

    _lib._lib = None
    assert not _lib.lib().is_emulator
    AF.invalidate_weight_cache()
    m = E2E(BC.ODIM, modality)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.load_state_dict(BC.bench_state_dict(m.state_dict(), seed))
    return m.cuda().train()


@pytest.mark.parametrize("tag", ["A", "B", "AA"])  # AA (round 4): the AUDIO model (BASELINE configs[3]) at batch A's geometry
@pytest.mark.parametrize("mode", ["precise", "hpf", "mixed", "bf16"])
def test_bench_shape_parity(gold, tag, mode):
    from auto_avsr_amd import functional as AF

    case = gold[tag]
    m = _model(case["seed"], case.get("modality", "video"))
    with AF.numerics(mode):
        r = BC.measure(m, case, torch.device("cuda"))
    AF.invalidate_weight_cache()
    print(f"\nPARITY batch {tag} ({len(case['lengths'])} x {max(case['lengths'])}) mode {mode}: " + json.dumps(
        {k: (round(v, 7) if isinstance(v, float) else v) for k, v in r.items()}))
    # the north-star bound in every mode: the three losses within 1e-3 relative
    assert r["loss_rel_err"] < 1e-3 and r["ctc_rel_err"] < 1e-3 and r["att_rel_err"] < 1e-3
    # round 5: WHOLE tensors (golden_bench_full_v1.pt: the reference's pred_pad (B, L+1, 5049), its raw CTC logits and encoder
    # output at 8 frames per utterance) -- plain relative L2, no selected columns, no log-softmax offset
    full, raw, enc = r["dec_logits_full_rel_l2"], r["ctc_logits_raw_rel_l2"], r["enc_full_rel_l2"]
    if mode in ("precise", "hpf"):
        assert full < 1e-4 and raw < 1e-4 and enc < 1e-4
    elif mode == "mixed":
        # THE BENCHMARKED MODE: the whole decoder-logit tensor inside the north star's 1e-3 with >= 20 % in hand at every
        # benchmarked shape (CPU model of the policy, tools/precision_study.py "mixed=f16a": 5.4e-4 / 6.3e-4 at A / B).  The
        # raw CTC logits and the encoder output carry the encoder's accumulated f16 activation rounding (~2e-3; the CTC LOSS,
        # which the bound names, is at 1e-5): bounded here so that a regression shows
        assert full < 8e-4, full
        assert raw < 3e-3 and enc < 3e-3
    else:
        assert full < 1.5e-2 and raw < 4e-2 and enc < 4e-2
    if mode == "precise":
        assert r["dec_logits_rel_l2"] < 1e-3 and r["ctc_logp_rel_l2"] < 1e-3 and r["enc_rel_l2"] < 1e-3
        assert r["acc"] == pytest.approx(r["acc_ref"], abs=1e-6) and r["acc_ref"] > 0.1
        assert r["grad_norm_rel_err_max"] < 1e-2 and r["grad_sample_cos_min"] > 0.999
        assert r["grad_sample_rel_l2_max"] < 2e-2, r.get("worst_sample_tensor")
    elif mode == "hpf":
        # the precise FORWARD (measured 1e-5 on the logits) with the bf16 backward (gradient samples: cosine 0.998, median
        # relative L2 7.5e-3 at batch A)
        assert r["dec_logits_rel_l2"] < 1e-4 and r["ctc_logp_rel_l2"] < 1e-4 and r["enc_rel_l2"] < 1e-3
        assert r["acc"] == pytest.approx(r["acc_ref"], abs=1e-6)
        assert r["grad_sample_cos_min"] > 0.99 and r["grad_sample_rel_l2_median"] < 2e-2 and r["grad_norm_rel_err_median"] < 5e-3
    elif mode == "mixed":
        # THE BENCHMARKED MODE (bench.py's headline): logits / CTC log-probabilities inside the north star's 1e-3 -- measured
        # on MI355X with the default policy (f16 encoder + trunk + decoder, split stem / projections / CTC head): 8.0e-4 /
        # 4.5e-4 at batch A (tools/r4_policy_sweep.sh sweep) -- with the bf16 backward
        assert r["dec_logits_rel_l2"] < 1e-3 and r["ctc_logp_rel_l2"] < 1e-3
        assert r["enc_rel_l2"] < 1.5e-2  # (a 32-channel slice of the encoder output: ~10x the logits' relative error in every mode)
        assert r["acc"] == pytest.approx(r["acc_ref"], abs=1e-6)
        assert r["grad_sample_cos_min"] > 0.99 and r["grad_sample_rel_l2_median"] < 3e-2 and r["grad_norm_rel_err_median"] < 5e-3
    else:
        # bf16 operands everywhere (8 significant bits): measured on MI355X decoder logits 7.4e-3, CTC log-probs 4.2e-3, encoder
        # slice 4e-2 at batch A (batch B within 1.5x) -- 7x / 4x OUTSIDE the north-star bound, which is why it is not the
        # benchmarked mode; the bounds below would catch a regression of ~1.5x
        assert r["dec_logits_rel_l2"] < 1.2e-2 and r["ctc_logp_rel_l2"] < 8e-3 and r["enc_rel_l2"] < 8e-2
        assert abs(r["acc"] - r["acc_ref"]) < 0.02
        assert r["grad_sample_cos_min"] > 0.9 and r["grad_sample_cos_mean"] > 0.99
        assert r["grad_norm_rel_err_median"] < 2e-2
