"""Training trajectories against the REFERENCE's loop (tests/golden/make_golden_trajectory.py: the reference's modules + torch's
AdamW(0.9, 0.98, wd 0.03) + per-step warm-up cosine + clip_grad_norm_(10), lightning.py:48-52,86-114, train.py:41, cosine.py) on
four small batches cycled -- the product side: this build's E2E + FusedAdamW (csrc/optim.hip), stepping exactly as train.py's
native loop does.

* dropout 0, 50 steps: the losses, the learning rate and the gradient norm follow the reference step by step -- tightly while
  the model is far from its data (first steps), with bounded drift once it memorises the four batches (loss 15.9 -> 0.09: the
  end of the run is chaotic in any arithmetic; torch itself does not reproduce it across thread counts);
* dropout on (the reference's rates), 200 steps: own counter-based generator vs torch's Philox -- means over windows agree."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import trajectory_common as TJ  # noqa: E402
from synth import synth_state_dict  # noqa: E402


def _run(dev, mode, steps, dropout, seed=1234):
    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.e2e import E2E
    from auto_avsr_amd.optim import FusedAdamW

    AF.invalidate_weight_cache()
    torch.manual_seed(0)
    m = E2E(TJ.ODIM, "video", adim=TJ.D, aheads=TJ.H, eunits=TJ.U, elayers=TJ.NENC, dunits=TJ.U, dlayers=TJ.NDEC)
    m.load_state_dict(synth_state_dict(m.state_dict(), TJ.SEED), strict=True)
    if not dropout:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    m = m.to(dev).train()
    opt = FusedAdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0, warmup_steps=TJ.WARMUP,
                     total_steps=TJ.TOTAL, cast_weights=dev.type == "cuda")
    AF.manual_seed(seed)
    seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    AF.set_seed_tensor(seed_dev)
    data = [tuple(t.to(dev) for t in TJ.batch(k)) for k in range(4)]
    rows = []
    try:
        with AF.numerics(mode):
            for s in range(steps):
                for p in m.parameters():
                    p.grad = None
                seed_dev.add_(1)
                AF.manual_seed(seed)
                AF.new_step()
                AF.refresh_weight_cache()
                loss, loss_ctc, loss_att, hits, ntok = m.forward_tensors(*data[s % 4])
                loss.backward()
                opt.step()
                rows.append(dict(loss=float(loss), loss_ctc=float(loss_ctc), loss_att=float(loss_att), acc=float(hits) / float(ntok),
                                 grad_norm=opt.last_grad_norm, lr=opt.last_lr))
    finally:
        AF.set_seed_tensor(None)
        AF.invalidate_weight_cache()
    return rows, m


def _fixture():
    return torch.load(TJ.FIXTURE, weights_only=False)


def test_first_steps_follow_reference(dev):
    """CPU suite (emulator) and GPU: three steps in the precise arithmetic -- loss terms, learning rate and gradient norm of the
    reference's loop to 1e-4 / 1e-3."""
    ref = _fixture()["nodrop"]["steps"]
    rows, _ = _run(dev, "precise", 3, False)
    for s, (a, b) in enumerate(zip(rows, ref)):
        for k in ("loss", "loss_ctc", "loss_att"):
            assert abs(a[k] - b[k]) <= 2e-4 * abs(b[k]), (s, k, a[k], b[k])
        assert abs(a["lr"] - b["lr"]) <= 1e-9 + 1e-6 * b["lr"], (s, a["lr"], b["lr"])
        assert abs(a["grad_norm"] - b["grad_norm"]) <= 2e-3 * b["grad_norm"], (s, a["grad_norm"], b["grad_norm"])
        assert abs(a["acc"] - b["acc"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["hpf", "mixed"])
def test_fifty_steps_follow_reference(mode):
    """The numerics train.py trains in (mixed) and round 3's tolerance-meeting mode (hpf): 50 steps, dropout 0."""
    ref = _fixture()["nodrop"]
    rows, m = _run(torch.device("cuda"), mode, 50, False)
    dev_rel = [abs(a["loss"] - b["loss"]) / max(abs(b["loss"]), 0.5) for a, b in zip(rows, ref["steps"])]
    print(f"\ntrajectory[{mode}] loss deviation (relative, floor 0.5): first 5 {['%.1e' % v for v in dev_rel[:5]]}, max over 50 "
          f"{max(dev_rel):.2e} at step {dev_rel.index(max(dev_rel))}; final loss {rows[-1]['loss']:.4f} (reference {ref['steps'][-1]['loss']:.4f})")
    # the first steps: the forward pass is inside 1e-3 of the reference's; from the second step on the loss also carries the
    # difference of the WEIGHTS after bf16-gradient updates (measured over seven runs: <= 9.5e-4 in the first five steps)
    assert dev_rel[0] < 1e-3, (rows[0], ref["steps"][0])
    for s in range(5):
        assert dev_rel[s] < 3e-3, (s, rows[s], ref["steps"][s])
        assert abs(rows[s]["grad_norm"] - ref["steps"][s]["grad_norm"]) < 3e-2 * ref["steps"][s]["grad_norm"]
    for s, (a, b) in enumerate(zip(rows, ref["steps"])):
        assert abs(a["lr"] - b["lr"]) <= 1e-9 + 1e-6 * b["lr"]
    # Bounded drift: the run goes where the reference's goes.  bf16 gradients perturb the path of a model that memorises four
    # batches (loss 15.9 -> 0.09 in 50 steps), and the tail is not even reproducible run to run (atomics commit in a
    # run-dependent order): over fifteen runs on five boxes (`tools/traj_spread.py`) the largest deviation was 0.2 - 2.3 of
    # max(loss, 0.5), somewhere between steps 19 and 49, the mean loss of the last ten steps 0.16 - 0.41 against the reference's
    # 0.16, while the first ten steps stayed within 1.8 %, the first twenty within 3.6 % on average, and the probed weights within a
    # cosine of 0.993 -- hpf, whose FORWARD pass is the reference's
    # to 1e-5, drifts as far as mixed: the drift is the backward pass's.  Asserted: the first twenty steps (loss 15.9 -> 1.7)
    # closely, and that the run converges like the reference's.
    print(f"mean deviation: first 20 steps {sum(dev_rel[:20]) / 20:.3e}, all 50 {sum(dev_rel) / 50:.3e}")
    assert max(dev_rel[:10]) < 0.05 and sum(dev_rel[:20]) / 20 < 0.1
    tail = rows[-10:]
    assert sum(r["loss"] for r in tail) / 10 < 2.0 and sum(r["acc"] for r in tail) / 10 > 0.8  # (reference: 0.2, acc 1.0)
    sd = m.state_dict()
    for k, v in ref["probe"].items():  # a handful of weights after 50 updates: same direction of travel
        if "running_" in k:
            continue
        got = sd[k].detach().flatten()[:16].float().cpu()
        cos = float(torch.dot(got, v) / (got.norm() * v.norm()))
        assert cos > 0.9, (k, cos, got, v)


@pytest.mark.gpu
def test_dropout_run_matches_reference_statistics():
    """200 steps with the reference's dropout rates.  Masks are not comparable (own generator), statistics are: the mean loss
    over windows of the run, and the run must converge like the reference's (which ends at 0.10 - 0.21 with acc 1.0)."""
    ref = _fixture()["drop"]["steps"]
    runs = [_run(torch.device("cuda"), "mixed", 200, True, seed=sd)[0] for sd in (11, 12)]
    for w0, w1 in ((0, 10), (10, 40), (40, 100), (100, 200)):
        r = sum(x["loss"] for x in ref[w0:w1]) / (w1 - w0)
        g = [sum(x["loss"] for x in run[w0:w1]) / (w1 - w0) for run in runs]
        print(f"\ndropout run, steps {w0}-{w1}: reference mean loss {r:.4f}, product {g[0]:.4f} / {g[1]:.4f}")
        tol = 0.1 if w1 <= 10 else 0.5  # later windows: different mask sequences send a memorising model down different paths
        assert all(abs(v - r) <= tol * r + 0.05 for v in g), (w0, w1, r, g)
    assert all(run[-1]["acc"] == 1.0 or sum(x["acc"] for x in run[-20:]) / 20 > 0.97 for run in runs)
    assert runs[0][5]["loss"] != runs[1][5]["loss"]  # (two seeds: two mask sequences)
