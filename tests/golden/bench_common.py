"""Inputs / weights / sampling shared by the bench-shape golden generator (make_golden_bench.py, runs the reference) and
its consumers (tests/test_bench_parity.py, bench.py's `parity` block).  No reference import here: this module travels to
the GPU box.  See make_golden_bench.py for what the fixture holds."""
import os

import torch

from synth import _gen, synth_state_dict

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden_bench_v1.pt")
FIXTURE_FULL = os.path.join(HERE, "golden_bench_full_v1.pt")  # round 5: whole tensors (make_golden_bench_full.py)
FAV = 17
ODIM = 5049
NSAMP = 64
BATCHES = {"A": dict(lengths=[400, 380, 360, 340], L=64, seed=21), "B": dict(lengths=[100] * 16, L=16, seed=22),
           # round 4: the audio model (BASELINE configs[3]: asr_trlrs3_base) at batch A's geometry -- 640 samples per frame
           "AA": dict(lengths=[400, 380, 360, 340], L=64, seed=23, modality="audio")}


def sample_index(name, numel, seed=0):
    return torch.randint(0, numel, (NSAMP,), generator=_gen("sample/" + name, seed))


def bench_state_dict(template, seed):
    sd = synth_state_dict(template, seed)
    sd["decoder.output_layer.bias"] = sd["decoder.output_layer.bias"].clone()
    sd["decoder.output_layer.bias"][FAV] += 6.0
    return sd


def bench_batch(lengths, L, seed, modality="video"):
    """x (B, Tmax, 1, 88, 88) [audio: (B, 640 Tmax, 1)] zero-padded, lengths in frames [audio: in samples], y (B, 1, L) with
    label length round(T / 6.25) capped at L, pad -1."""
    g = torch.Generator().manual_seed(5000 + seed)
    B, T = len(lengths), max(lengths)
    x = torch.zeros(B, T, 1, 88, 88) if modality == "video" else torch.zeros(B, 640 * T, 1)
    y = torch.full((B, 1, L), -1, dtype=torch.int64)
    for b, t in enumerate(lengths):
        if modality == "video":
            x[b, :t] = torch.randn(t, 1, 88, 88, generator=g)
        else:
            x[b, : 640 * t] = torch.randn(640 * t, 1, generator=g)
        n = min(L, max(1, round(t / 6.25)))
        lab = torch.randint(1, ODIM - 1, (n,), generator=g)
        lab[::5] = FAV
        y[b, 0, :n] = lab
    return x, torch.tensor(lengths, dtype=torch.int64) * (1 if modality == "video" else 640), y


def raw_frames(nfr):
    """Frames at which the whole-tensor fixture keeps the raw CTC logits / the encoder output of every utterance."""
    return torch.linspace(0, nfr - 1, 8).round().long()


def load_full(tag):
    """The whole-tensor reference outputs of case `tag` (None when the fixture is absent)."""
    if not os.path.exists(FIXTURE_FULL):
        return None
    return torch.load(FIXTURE_FULL, weights_only=False)[tag]


def full_errors(full, dec, ctc, enc):
    """Unflattered forward errors: relative L2 over the WHOLE decoder-logit tensor `pred_pad` (B, L+1, V), over the RAW CTC
    logits and over the encoder output at the fixture's frames.  `dec` (B, L+1, >=V), `ctc` (B, T, >=V), `enc` (B, T, D)."""
    ts = full["tsel_raw"]
    return {"dec_logits_full_rel_l2": rel(dec.float().cpu()[..., :ODIM], full["pred_pad"]),
            "ctc_logits_raw_rel_l2": rel(ctc.float().cpu()[:, ts][..., :ODIM], full["ys_hat"]),
            "enc_full_rel_l2": rel(enc.float().cpu()[:, ts], full["enc"])}


def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def measure(model, case, device):
    """Run `model` (an auto_avsr_amd E2E carrying bench_state_dict weights, training mode, dropout 0) on the case's batch in
    the CURRENT numerical mode and compare with the reference numbers in `case`.  Returns a dict of measured errors:
    relative errors of the three losses, accuracy difference, relative L2 error of the decoder-logit / CTC log-prob /
    encoder-output slices, and over all parameter gradients: worst / median relative norm error, worst / mean cosine and
    worst relative L2 error of the 64 sampled elements per tensor (tensors whose reference gradient is numerically zero
    are skipped)."""
    import statistics

    x, lengths, y = bench_batch(case["lengths"], case["L"], case["seed"], case.get("modality", "video"))
    grab = {}
    hooks = [model.encoder.register_forward_hook(lambda m, i, o: grab.__setitem__("enc", o[0].detach())),
             model.decoder.register_forward_hook(lambda m, i, o: grab.__setitem__("dec", o[0].detach())),
             # CTC.forward returns (loss, logits (T, B, V)) on both sides (ctc.py:62-65); this build never calls ctc_lo itself
             model.ctc.register_forward_hook(lambda m, i, o: grab.__setitem__("ctc", o[1].detach().transpose(0, 1)))]
    for p in model.parameters():
        p.grad = None
    try:
        loss, loss_ctc, loss_att, acc = model(x.to(device), lengths.to(device), y.to(device))
        loss.backward()
    finally:
        for h in hooks:
            h.remove()
    out = {"loss_rel_err": abs(float(loss) - case["loss"]) / abs(case["loss"]),
           "ctc_rel_err": abs(float(loss_ctc) - case["loss_ctc"]) / abs(case["loss_ctc"]),
           "att_rel_err": abs(float(loss_att) - case["loss_att"]) / abs(case["loss_att"]),
           "acc": float(acc), "acc_ref": case["acc"]}
    vcols, tsel = case["vcols"], case["tsel"]
    dec = grab["dec"].float().cpu()[..., : ODIM][:, :, vcols]
    out["dec_logits_rel_l2"] = rel(dec, case["dec_logits"])
    ctc = grab["ctc"].float().cpu()[..., : ODIM]
    ctc_logp = torch.log_softmax(ctc, -1)[:, tsel][:, :, vcols]
    out["ctc_logp_rel_l2"] = rel(ctc_logp, case["ctc_logp"])
    out["enc_rel_l2"] = rel(grab["enc"].float().cpu()[:, tsel, :32], case["enc"])
    full = load_full(case["tag"])
    if full is not None:
        out.update(full_errors(full, grab["dec"], grab["ctc"], grab["enc"]))
    gmax = max(case["grad_norms"].values())
    norm_err, cos, samp = [], [], []
    worst = {}
    for k, p in model.named_parameters():
        ref_n = case["grad_norms"][k]
        if ref_n < 1e-6 * gmax:
            continue
        g = p.grad.detach().float().cpu().reshape(-1)
        ne = abs(float(g.double().norm()) - ref_n) / ref_n
        got_s, ref_s = g[sample_index(k, g.numel())].double(), case["grad_samples"][k].double()
        c = float(torch.dot(got_s, ref_s) / (got_s.norm() * ref_s.norm() + 1e-300))
        se = float((got_s - ref_s).norm() / ref_s.norm().clamp_min(1e-300))
        norm_err.append(ne)
        cos.append(c)
        samp.append(se)
        if se >= max(samp):
            worst["worst_sample_tensor"] = k
        if c <= min(cos):
            worst["worst_cos_tensor"] = f"{k} (reference norm {ref_n / gmax:.1e} of the largest)"
    out.update(grad_tensors=len(norm_err), grad_norm_rel_err_max=max(norm_err),
               grad_norm_rel_err_median=statistics.median(norm_err), grad_sample_cos_min=min(cos),
               grad_sample_cos_mean=sum(cos) / len(cos), grad_sample_rel_l2_max=max(samp),
               grad_sample_rel_l2_median=statistics.median(samp), **worst)
    return out
