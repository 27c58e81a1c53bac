"""Golden vectors at the BENCHMARKED shapes (batch A = 4 x 400 frames / 64 labels with lengths
400/380/360/340, batch B = 16 x 100 / 16), full-size video model, produced by running the REFERENCE implementation
(/root/reference, PyTorch CPU fp32).  Run in the build container only:

    python tests/golden/make_golden_bench.py      ->  tests/golden/golden_bench_v1.pt

Stored per batch: the three losses and the accuracy, a slice of the decoder logits and of the CTC log-probabilities, a
slice of the encoder output, every parameter-gradient norm AND 64 sampled elements of every gradient tensor (indices from
`sample_index`, a pure function of the tensor name -- the test regenerates them).  Weights come from synth.py on both
sides; on top of them `decoder.output_layer.bias[FAV] += 6`, and every fifth label is FAV: the decoder's arg-max is then
FAV at almost every position, so the teacher-forced accuracy is ~0.2 instead of the 0.0 random weights give (a non-vacuous
check of th_accuracy's masking / counting).  Dropout probabilities are 0; BatchNorm in training mode."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)

from espnet.nets.pytorch_backend.e2e_asr_conformer import E2E  # noqa: E402

from bench_common import BATCHES, FAV, FIXTURE, ODIM, bench_batch, bench_state_dict, sample_index  # noqa: E402


def case(tag):
    cfg = BATCHES[tag]
    modality = cfg.get("modality", "video")
    torch.manual_seed(0)
    m = E2E(ODIM, modality)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.load_state_dict(bench_state_dict(m.state_dict(), cfg["seed"]))
    m.train()
    x, lengths, y = bench_batch(cfg["lengths"], cfg["L"], cfg["seed"], modality)
    grab = {}
    m.encoder.register_forward_hook(lambda mod, i, o: grab.__setitem__("enc", o[0].detach()))
    m.decoder.register_forward_hook(lambda mod, i, o: grab.__setitem__("dec", o[0].detach()))
    m.ctc.ctc_lo.register_forward_hook(lambda mod, i, o: grab.__setitem__("ctc", o.detach()))
    t0 = time.time()
    loss, loss_ctc, loss_att, acc = m(x, lengths, y)
    loss.backward()
    dt = time.time() - t0
    vcols = torch.cat([torch.tensor([0, FAV, ODIM - 1]), sample_index("vocab", ODIM)[:29]])
    ctc_logp = torch.log_softmax(grab["ctc"], -1)  # (B, T, V)
    nfr = max(cfg["lengths"])
    tsel = torch.arange(0, nfr, max(1, nfr // 25))
    out = dict(tag=tag, lengths=cfg["lengths"], L=cfg["L"], seed=cfg["seed"], modality=modality, loss=float(loss), loss_ctc=float(loss_ctc),
               loss_att=float(loss_att), acc=float(acc), seconds=dt, vcols=vcols, tsel=tsel,
               dec_logits=grab["dec"][:, :, vcols].clone(), ctc_logp=ctc_logp[:, tsel][:, :, vcols].clone(),
               enc=grab["enc"][:, tsel, :32].clone(),
               grad_norms={k: float(p.grad.double().norm()) for k, p in m.named_parameters()},
               grad_samples={k: p.grad.reshape(-1)[sample_index(k, p.numel())].clone() for k, p in m.named_parameters()})
    print(tag, out["loss"], out["loss_ctc"], out["loss_att"], out["acc"], f"{dt:.1f}s", flush=True)
    return out


if __name__ == "__main__":
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    # cases already in the fixture are kept as they are (round 4 added "AA"; A / B are the round-2 reference runs)
    res = torch.load(FIXTURE, weights_only=False) if os.path.exists(FIXTURE) else {"torch_version": torch.__version__, "FAV": FAV}
    for tag in BATCHES:
        if tag not in res:
            res[tag] = case(tag)
    torch.save(res, FIXTURE)
