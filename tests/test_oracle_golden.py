"""Pin the oracle (oracle/avsr_oracle.py) against outputs of the reference implementation recorded by
tests/golden/make_golden.py.  CPU only; this is what makes the oracle trustworthy as the GPU checker."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
from synth import synth_batch, synth_tensor  # noqa: E402

import avsr_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def golden():
    return torch.load(os.path.join(HERE, "golden", "golden_v1.pt"), weights_only=False)


def _sd(shapes, seed, requires_grad=True):
    sd = {}
    for k, shp in shapes.items():
        dtype = torch.int64 if k.endswith("num_batches_tracked") else torch.float32
        t = synth_tensor(k, shp, dtype, seed)
        if requires_grad and t.is_floating_point() and "running_" not in k:
            t.requires_grad_()
        sd[k] = t
    return sd


def _check_grad_norms(sd, ref_norms, rtol=2e-3):
    bad = []
    # gradients that are analytically zero (a bias in front of a train-mode BatchNorm, linear_k.bias under the
    # softmax shift invariance) are pure rounding noise on both sides: compare those against an absolute floor
    atol = 1e-5 * max(ref_norms.values())
    for k, n in ref_norms.items():
        got = float(sd[k].grad.double().norm()) if sd[k].grad is not None else 0.0
        if abs(got - n) > rtol * n + atol:
            bad.append((k, got, n))
    assert not bad, bad[:5]


def test_encoder_small(golden):
    c = golden["encoder_small"]
    sd = _sd(c["shapes"], c["seed"])
    x = c["x"].clone().requires_grad_()
    mask = (torch.arange(x.shape[1])[None] < c["lengths"][:, None]).unsqueeze(-2)
    out = O.conformer_encoder(sd, "", x, mask, H=2, train_bn=True)
    assert (out - c["out"]).abs().max() < 2e-5
    (out * c["w"]).sum().backward()
    assert (x.grad - c["dx"]).abs().max() < 1e-4 * c["dx"].abs().max()
    _check_grad_norms(sd, c["grad_norms"])


def test_decoder_small(golden):
    c = golden["decoder_small"]
    sd = _sd(c["shapes"], c["seed"])
    mem = c["memory"].clone().requires_grad_()
    L = c["ys_in"].shape[1]
    mmask = (torch.arange(mem.shape[1])[None] < c["lengths"][:, None]).unsqueeze(-2)
    tmask = (c["ys_in"] != -1).unsqueeze(-2) & torch.tril(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0)
    out = O.transformer_decoder(sd, "", c["ys_in"], tmask, mem, mmask, H=2)
    assert (out - c["out"]).abs().max() < 2e-5
    (out * c["w"]).sum().backward()
    assert (mem.grad - c["dmemory"]).abs().max() < 1e-4 * c["dmemory"].abs().max()
    _check_grad_norms(sd, c["grad_norms"])


@pytest.mark.parametrize("name", ["e2e_video", "e2e_audio"])
def test_e2e(golden, name):
    c = golden[name]
    sd = _sd(c["shapes"], c["seed"])
    x, lengths, y = synth_batch(c["modality"], c["B"], c["T"], c["L"], 5049, c["seed"])
    (loss, loss_ctc, loss_att, acc), aux = O.e2e_forward(sd, x, lengths, y, modality=c["modality"])
    assert (aux["feats"][:, :, :16] - c["feats_sample"]).abs().max() < 1e-4 * c["feats_sample"].abs().max()
    for got, key in ((loss, "loss"), (loss_ctc, "loss_ctc"), (loss_att, "loss_att")):
        assert abs(float(got) - c[key]) < 1e-4 * abs(c[key]), (key, float(got), c[key])
    assert acc == c["acc"]
    loss.backward()
    _check_grad_norms(sd, c["grad_norms"], rtol=5e-3)


def test_e2e_at_bench_shape():
    """The oracle at the BENCHMARKED shape: the survey's batch B (16 x 100 frames, 16 labels, full-size video model)
    against the reference numbers of tests/golden/make_golden_bench.py -- losses, accuracy (non-zero by construction) and
    64 sampled elements of every parameter gradient, not just the norms."""
    import bench_common as BC
    from auto_avsr_amd.e2e import E2E

    c = torch.load(BC.FIXTURE, weights_only=False)["B"]
    shapes = {k: tuple(v.shape) for k, v in E2E(BC.ODIM, "video").state_dict().items()}
    sd = BC.bench_state_dict({k: torch.empty(s, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
                              for k, s in shapes.items()}, c["seed"])
    sd = {k: (v.requires_grad_() if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}
    x, lengths, y = BC.bench_batch(c["lengths"], c["L"], c["seed"])
    torch.set_num_threads(8)
    (loss, loss_ctc, loss_att, acc), _ = O.e2e_forward(sd, x, lengths, y, modality="video")
    for got, key in ((loss, "loss"), (loss_ctc, "loss_ctc"), (loss_att, "loss_att")):
        assert abs(float(got) - c[key]) < 1e-4 * abs(c[key]), (key, float(got), c[key])
    assert abs(float(acc) - c["acc"]) < 1e-9 and c["acc"] > 0.1
    loss.backward()
    gmax = max(c["grad_norms"].values())
    bad = []
    for k, ref in c["grad_samples"].items():
        if c["grad_norms"][k] < 1e-6 * gmax:
            continue
        got = sd[k].grad.reshape(-1)[BC.sample_index(k, sd[k].numel())]
        err = float((got.double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-300))
        if err > 5e-3:
            bad.append((k, err))
    assert not bad, bad[:5]


def test_e2e_audio_at_bench_shape_forward():
    """The oracle on the AUDIO model (BASELINE configs[3]) at batch A's geometry (4 x 400 frames = 256 000 samples), forward only:
    the three losses, the accuracy and the decoder-logit / CTC log-probability slices of the reference run (fixture case "AA",
    tests/golden/make_golden_bench.py)."""
    import bench_common as BC
    from auto_avsr_amd.e2e import E2E

    c = torch.load(BC.FIXTURE, weights_only=False)["AA"]
    assert c["modality"] == "audio"
    shapes = {k: tuple(v.shape) for k, v in E2E(BC.ODIM, "audio").state_dict().items()}
    sd = BC.bench_state_dict({k: torch.empty(s, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
                              for k, s in shapes.items()}, c["seed"])
    x, lengths, y = BC.bench_batch(c["lengths"], c["L"], c["seed"], "audio")
    torch.set_num_threads(8)
    with torch.no_grad():
        (loss, loss_ctc, loss_att, acc), mid = O.e2e_forward(sd, x, lengths, y, modality="audio")
        ctc = O.linear(sd, "ctc.ctc_lo.", mid["enc"])
    for got, key in ((loss, "loss"), (loss_ctc, "loss_ctc"), (loss_att, "loss_att")):
        assert abs(float(got) - c[key]) < 1e-4 * abs(c[key]), (key, float(got), c[key])
    assert abs(float(acc) - c["acc"]) < 1e-9 and c["acc"] > 0.1
    assert BC.rel(mid["pred"][:, :, c["vcols"]], c["dec_logits"]) < 1e-4
    assert BC.rel(torch.log_softmax(ctc, -1)[:, c["tsel"]][:, :, c["vcols"]], c["ctc_logp"]) < 1e-4
