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
        # THE BENCHMARKED MODE: the whole decoder-logit tensor inside the north star's 1e-3 at every benchmarked shape.  The figure
        # is a sum of ~10^2 independent f16 rounding contributions and RE-ROLLS by up to ~10 % with any change of summation order
        # upstream (round 6: a different tile shape in the first ResNet stage -- bit-compatible MFMA order, BatchNorm partial sums
        # grouped by 256 instead of 128 rows -- moved batch B from 7.34e-4 to 7.65e-4 and batch A from 5.3e-4 to 6.0e-4 with no
        # change of arithmetic).  So the regression tripwire is the MEDIAN of three such rolls (the default tiles and two
        # alternative tile choices of the front-end, avsr_tune knobs 21 / 18) < 8e-4, and EVERY roll must be inside the north
        # star's 1e-3.  The raw CTC logits and the encoder output carry the encoder's accumulated f16 activation rounding (measured
        # 1.9e-3 / 1.7e-3 / 2.0e-3 at A / B / AA; the CTC LOSS, which the bound names, is at 1e-5): bounded at measured + 15 %
        from auto_avsr_amd import ops

        rolls = [full]
        for knob, val in ((21, 23), (18, 7)):
            ops.tune(knob, val)
            try:
                mm = _model(case["seed"], case.get("modality", "video"))
                with AF.numerics(mode):
                    rolls.append(BC.measure(mm, case, torch.device("cuda"))["dec_logits_full_rel_l2"])
            finally:
                ops.tune(knob, 0)
                AF.invalidate_weight_cache()
        print(f"PARITY batch {tag} mixed: dec_logits_full_rel_l2 over three summation orders: {[round(v, 7) for v in rolls]}")
        assert max(rolls) < 1e-3, rolls
        assert sorted(rolls)[1] < 8e-4, rolls
        assert raw < 2.3e-3 and enc < 2.3e-3, (raw, enc)
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
        # gradient tolerance of the benchmarked mode (bf16 backward on bf16 twins of the f16 / split-plane forward; stated in
        # DESIGN.md section 2): per tensor, 64 sampled elements -- median relative L2 <= 2e-2 (measured 1.1e-2 / 1.4e-2 at A / B),
        # cosine >= 0.99 on every tensor whose gradient is not numerically zero, median norm error <= 5e-3
        assert r["grad_sample_cos_min"] > 0.99, r.get("worst_cos_tensor")
        assert r["grad_sample_rel_l2_median"] < 2e-2 and r["grad_norm_rel_err_median"] < 5e-3
    else:
        # bf16 operands everywhere (8 significant bits): measured on MI355X decoder logits 7.4e-3, CTC log-probs 4.2e-3, encoder
        # slice 4e-2 at batch A (batch B within 1.5x) -- 7x / 4x OUTSIDE the north-star bound, which is why it is not the
        # benchmarked mode; the bounds below would catch a regression of ~1.5x
        assert r["dec_logits_rel_l2"] < 1.2e-2 and r["ctc_logp_rel_l2"] < 8e-3 and r["enc_rel_l2"] < 8e-2
        assert abs(r["acc"] - r["acc_ref"]) < 0.02
        assert r["grad_sample_cos_min"] > 0.9 and r["grad_sample_cos_mean"] > 0.99
        assert r["grad_norm_rel_err_median"] < 2e-2
