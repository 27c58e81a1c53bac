"""Deterministic mode (/root/reference/train.py:18 `seed_everything(42)`: the reference's CPU path is
reproducible run to run).  The default build forms several parameter-gradient sums with floating-point atomics from many blocks
(split-K weight gradients, bias / LayerNorm / depthwise-convolution parameter gradients, position-bias and position-projection
gradients), which commit in a run-dependent order: two runs of the same steps differ in the last bits, and Adam amplifies that.
`functional.set_deterministic(True)` (`train.py --deterministic`, `AVSR_DETERMINISTIC=1`) forms every such sum in a fixed order:
two runs of the native loop -- hipGraph replay on the GPU, the multi-threaded emulator on the CPU -- give bit-identical losses and
bit-identical weights, with dropout on (the masks are counter-based and reproducible by construction)."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(dev, tmp, det, steps_epochs, monkeypatch):
    import auto_avsr_amd.synthetic as S
    from auto_avsr_amd import functional as AF
    from auto_avsr_amd import train_native as TN
    from auto_avsr_amd.e2e import E2E

    lengths = [12, 14, 20, 22, 30, 33, 12, 21] if dev.type == "cuda" else [3, 4, 5, 3]
    monkeypatch.setattr(S, "utterance_lengths", lambda n=6, seed=42, lo=12, hi=400: torch.tensor(lengths).numpy())
    AF.invalidate_weight_cache()
    torch.manual_seed(0)
    m = E2E(30, "video", adim=128, aheads=2, eunits=64, elayers=2, dunits=64, dlayers=1, cnn_module_kernel=7).to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.1
    args = types.SimpleNamespace(modality="video", max_frames=48 if dev.type == "cuda" else 8, train_num_buckets=3, lr=1e-3,
                                 weight_decay=0.03, warmup_epochs=1, max_epochs=steps_epochs, exp_dir=None, exp_name="run", ckpt_path=None,
                                 steps=None, val_batches=0, synthetic_utterances=len(lengths), log_every=1,
                                 numerics="mixed" if dev.type == "cuda" else "precise", synthetic=True, deterministic=det)
    t0 = time.perf_counter()
    losses = TN.fit(m, args, dev, log=lambda s: None)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert not AF.deterministic()  # the loop put the switch back
    w = torch.cat([p.detach().float().flatten().cpu() for p in m.parameters()] + [b.detach().float().flatten().cpu() for b in m.buffers()])
    AF.invalidate_weight_cache()
    return losses, w, dt, TN.fit.last_stats


def test_two_runs_bit_identical(dev, tmp_path, monkeypatch):
    epochs = 5 if dev.type == "cuda" else 2
    a = _run(dev, tmp_path, True, epochs, monkeypatch)
    b = _run(dev, tmp_path, True, epochs, monkeypatch)
    assert len(a[0]) >= (10 if dev.type == "cuda" else 2) and all(v == v for v in a[0])
    assert a[0] == b[0], (a[0], b[0])            # losses: equal as python floats
    assert torch.equal(a[1], b[1])               # weights AND BatchNorm buffers: bit for bit
    if dev.type == "cuda":
        assert a[3]["replayed"] > 0               # ... under hipGraph replay
    if dev.type != "cuda":
        return  # (the third run costs the emulator another 50 s; the comparison below is about the GPU's atomics)
    # the default mode trains alike (same arithmetic up to summation order) -- and is free to differ in the last bits
    c = _run(dev, tmp_path, False, epochs, monkeypatch)
    assert len(c[0]) == len(a[0])
    for x, y in zip(a[0][:3], c[0][:3]):
        assert abs(x - y) <= 2e-2 * abs(x), (a[0], c[0])
    print(f"\ndeterministic {a[2]:.2f} s / {b[2]:.2f} s, default {c[2]:.2f} s; default == deterministic bitwise: {torch.equal(a[1], c[1])}")


def test_ctc_branch_on_a_second_stream_same_bits(dev, tmp_path, monkeypatch):
    """Round 6: E2E.forward_tensors issues the CTC branch (ctc.py:32-38) on a second stream beside the decoder
    (functional._SIDE_BRANCH); autograd runs the branch's backward there too and the hipGraph capture turns fork / join into graph
    edges.  Same kernels on the same operands, only concurrent: in the deterministic mode the losses and the trained weights are
    bit-identical with the branch on one stream or two."""
    import pytest

    from auto_avsr_amd import functional as AF

    if dev.type != "cuda":
        pytest.skip("streams: GPU only")
    epochs = 4
    monkeypatch.setattr(AF, "_SIDE_BRANCH", True)
    a = _run(dev, tmp_path, True, epochs, monkeypatch)
    monkeypatch.setattr(AF, "_SIDE_BRANCH", False)
    b = _run(dev, tmp_path, True, epochs, monkeypatch)
    assert a[3]["replayed"] > 0 and len(a[0]) >= 8
    assert a[0] == b[0], (a[0], b[0])
    assert torch.equal(a[1], b[1])
