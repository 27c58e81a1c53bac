"""Dropout of the training hot path is statistically, not bitwise, comparable with the reference's (torch's Philox stream cannot
be matched): the kernels draw every mask from a stateless hash of (site seed, device-side step counter, element
index) -- csrc/prims.h dropout_scale -- so that the backward pass regenerates the forward mask and a replayed hipGraph gets a new
mask per step.  This file checks the generator where it matters: at EVERY dropout site of a model's forward pass (the reference
has 63 modules with p = 0.1: conformer_encoder.py:111-162, transformer_decoder.py:65-128, embedding.py, positionwise_feed_forward.py)
the keep rate is 1 - p within 4 sigma, kept elements carry 1 / (1 - p), masks of different sites and of different steps are
independent, and the same (site, step) gives the same mask again."""
import math
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _sites(dev, full):
    """(p, seed) of every dropout site one training-mode forward pass visits, in visiting order."""
    from synth import synth_batch

    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.e2e import E2E

    torch.manual_seed(0)
    kw = {} if full else dict(adim=128, aheads=2, eunits=128, elayers=2, dunits=128, dlayers=2, cnn_module_kernel=7)
    odim = 5049 if full else 40
    m = E2E(odim, "video", **kw).to(dev).train()
    seen = []
    orig = AF._drop_args

    def spy(p, ref):
        out = orig(p, ref)
        if out[0] > 0.0:
            seen.append((out[0], out[1]))
        return out

    x, lens, y = synth_batch("video", 2, 6, 3, odim, seed=1)
    mods = [AF] + [sys.modules[n] for n in list(sys.modules) if n.startswith("auto_avsr_amd.functional_")]
    saved = [(mod, mod._drop_args) for mod in mods if hasattr(mod, "_drop_args")]
    for mod, _ in saved:
        mod._drop_args = spy
    try:
        AF.invalidate_weight_cache()
        AF.manual_seed(77)
        AF.new_step()
        with torch.no_grad():
            m.forward_tensors(x.to(dev), lens.to(dev), y.to(dev))
    finally:
        for mod, fn in saved:
            mod._drop_args = fn
        AF.invalidate_weight_cache()
    n_layers = (12, 6) if full else (2, 2)
    return seen, n_layers


def _check(dev, seen, n):
    from auto_avsr_amd import ops

    ones = torch.ones(n, device=dev)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    masks = []
    for p, seed in seen:
        out = ops.scale_dropout(ones, torch.float32, drop_p=p, seed=seed, seed_dev=step)
        kept = out != 0
        assert torch.allclose(out[kept], torch.full((1,), 1.0 / (1.0 - p), device=dev)), "kept elements are scaled by 1 / (1 - p)"
        rate = float(kept.float().mean())
        sigma = math.sqrt(p * (1 - p) / n)
        assert abs(rate - (1 - p)) < 4 * sigma, (seed, rate, 1 - p, sigma)
        masks.append(kept)
    assert len({s for _, s in seen}) == len(seen), "every site draws from its own seed"
    for a, b, (p, _), (q, _) in zip(masks, masks[1:], seen, seen[1:]):  # neighbouring sites: independent masks
        both = float((a & b).float().mean())
        want = (1 - p) * (1 - q)
        assert abs(both - want) < 4 * math.sqrt(want * (1 - want) / n), (both, want)
    # the device-side step counter: a new mask per step, the same mask for the same step (what the backward pass relies on)
    p, seed = seen[0]
    again = ops.scale_dropout(ones, torch.float32, drop_p=p, seed=seed, seed_dev=step) != 0
    step.add_(1)
    nxt = ops.scale_dropout(ones, torch.float32, drop_p=p, seed=seed, seed_dev=step) != 0
    assert torch.equal(again, masks[0])
    both = float((nxt & masks[0]).float().mean())
    assert abs(both - (1 - p) ** 2) < 4 * math.sqrt((1 - p) ** 2 * (1 - (1 - p) ** 2) / n)


def test_dropout_sites_small_model(dev):
    seen, (ne, nd) = _sites(dev, full=False)
    print(f"\n{len(seen)} dropout sites with p > 0 in one forward pass of a {ne} + {nd} layer model")
    # per encoder layer: 2 x (FFN inner + FFN residual) + attention residual + convolution residual; per decoder layer: FFN inner +
    # 3 residuals + 2 attention-probability dropouts; + the two positional-encoding dropouts (encoder input, decoder embedding)
    assert 6 * ne + 6 * nd + 2 <= len(seen) <= 6 * ne + 6 * nd + 4
    assert all(abs(p - 0.1) < 1e-9 for p, _ in seen)
    _check(dev, seen, 1 << 16)


@pytest.mark.gpu
def test_dropout_sites_full_model():
    seen, _ = _sites(torch.device("cuda"), full=True)
    print(f"\n{len(seen)} dropout sites with p > 0 in one forward pass of the full-size model (reference: 63 MODULES with p = 0.1, "
          "several of them applied more than once per layer: 6 applications per encoder layer, 6 per decoder layer, 2 positional encodings, the CTC head)")
    assert 12 * 6 + 6 * 6 + 2 <= len(seen) <= 12 * 6 + 6 * 6 + 4
    _check(torch.device("cuda"), seen, 1600 * 768)
