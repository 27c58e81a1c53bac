"""The Lightning `Trainer` path (/root/reference/train.py:30-42,199-207 + lightning.py:48-52,86-114) without pytorch_lightning
(absent from the image): a stand-in trainer that drives `lightning.ModelModule` through the hook protocol a Lightning fit loop
follows -- `on_fit_start` -> `configure_optimizers` -> per batch {`training_step` -> `backward` -> clip at
`gradient_clip_val` -> `optimizer.step` -> `zero_grad` -> `lr_scheduler.step` (interval "step")} under AUTOMATIC optimisation,
per batch {`training_step`} alone under MANUAL optimisation -> `on_fit_end`.

Two modes of ModelModule:
* `--trainer-step auto` (the reference's protocol): eager launches, torch.optim.AdamW + cosine.WarmupCosineScheduler, Trainer-side
  clipping;
* `--trainer-step native` (round 6): manual optimisation; `training_step` runs the whole fused step through
  `auto_avsr_amd.train_native.NativeStepper` -- one replayed hipGraph per batch shape on the GPU, the step bench.py times.

The two must train alike (same AdamW / clip / schedule arithmetic), and the native mode must replay graphs on the GPU."""
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StandInTrainer:
    """What ModelModule sees of a Lightning Trainer (attributes: datamodule, num_devices, num_nodes, global_rank) and the order in
    which a fit loop calls into the module."""

    def __init__(self, datamodule, max_epochs, gradient_clip_val=10.0):
        self.datamodule, self.max_epochs, self.gradient_clip_val = datamodule, max_epochs, gradient_clip_val
        self.num_devices = self.num_nodes = 1
        self.global_rank = 0
        self.losses, self.lrs = [], []

    def fit(self, module):
        module.trainer = self
        module.train()
        module.on_fit_start()
        try:
            conf = module.configure_optimizers()
            automatic = getattr(module, "automatic_optimization", True)
            if automatic:
                (opt,), (sch,) = conf
                assert sch["interval"] == "step"
                sched = sch["scheduler"]
            else:
                assert conf is None  # manual optimisation without Lightning-visible optimizers
                assert self.gradient_clip_val is None, "Lightning refuses gradient_clip_val under manual optimisation"
            for _ in range(self.max_epochs):
                for i, batch in enumerate(self.datamodule.train_dataloader()):
                    loss = module.training_step(batch, i)
                    if automatic:
                        loss.backward()
                        torch.nn.utils.clip_grad_norm_(module.parameters(), self.gradient_clip_val)
                        self.lrs.append(opt.param_groups[0]["lr"])  # the rate this update uses
                        opt.step()
                        opt.zero_grad(set_to_none=True)
                        sched.step()
                    self.losses.append(float(loss.detach()))
        finally:
            module.on_fit_end()
        return self.losses


class _Data:
    """train_dataloader() of a DataModule: a list of collated batches in the reference's dict layout (data_module.py:10-41)."""

    def __init__(self, batches):
        self.batches = batches

    def train_dataloader(self):
        return self.batches


def _batches(dev, odim, shapes, seed=3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for (B, T, L) in shapes:
        out.append({"inputs": torch.randn(B, T, 1, 88, 88, generator=g).to(dev),
                    "input_lengths": torch.full((B,), T, dtype=torch.int64).to(dev),
                    "targets": torch.randint(1, odim - 1, (B, L), generator=g).to(dev)})
    return out


def _module(dev, trainer_step, monkeypatch, numerics, max_epochs=3):
    import lightning as LM
    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.e2e import E2E

    odim = 30

    class Text:
        token_list = ["<blank>"] + [f"w{i}" for i in range(odim - 2)] + ["<eos>"]

    def small(n, modality, ctc_weight=0.1):
        torch.manual_seed(0)
        m = E2E(n, modality, ctc_weight=ctc_weight, adim=128, aheads=2, eunits=64, elayers=1, dunits=64, dlayers=1, cnn_module_kernel=7)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return m

    monkeypatch.setattr(LM, "TextTransform", Text)
    monkeypatch.setattr(LM, "E2E", small)
    AF.invalidate_weight_cache()
    args = types.SimpleNamespace(modality="video", lr=1e-3, weight_decay=0.03, warmup_epochs=1, max_epochs=max_epochs, ctc_weight=0.1,
                                 numerics=numerics, trainer_step=trainer_step, pretrained_model_path=None)
    return LM.ModelModule(args).to(dev), odim


def test_trainer_protocol_auto_and_native_train_alike(dev, monkeypatch):
    from auto_avsr_amd import functional as AF

    numerics = "precise" if dev.type == "cpu" else "mixed"
    shapes = [(2, 6, 3), (1, 8, 4)]
    epochs = 2 if dev.type == "cpu" else 3  # (the emulator runs the video front end at ~10 s per step)
    runs = {}
    for mode in ("auto", "native"):
        mod, odim = _module(dev, mode, monkeypatch, numerics, max_epochs=epochs)
        data = _Data(_batches(dev, odim, shapes))
        tr = StandInTrainer(data, max_epochs=epochs, gradient_clip_val=10.0 if mode == "auto" else None)
        before = AF.mode()
        losses = tr.fit(mod)
        assert AF.mode() == before and AF._state["bn_sync"] is None  # on_fit_end put everything back
        stats = None
        runs[mode] = (losses, tr.lrs, stats)
        if mode == "native":
            assert mod._native is None  # closed by on_fit_end
    la, ln = runs["auto"][0], runs["native"][0]
    n = 2 * epochs
    assert len(la) == len(ln) == n and all(l == l for l in la + ln)
    # the same training: torch AdamW + clip_grad_norm_ + WarmupCosineScheduler against the fused device-side step.  The first loss
    # is the same forward pass; later ones agree while Adam's sign-like first updates have not yet amplified rounding differences
    tol = 2e-3 if numerics == "precise" else 3e-2
    assert abs(la[0] - ln[0]) <= 1e-4 * abs(la[0]) + (0 if numerics == "precise" else 1e-2 * abs(la[0])), (la, ln)
    for a, b in zip(la, ln):
        assert abs(a - b) <= tol * abs(a), (la, ln)
    assert la[-2] < la[0] and ln[-2] < ln[0]  # (the same batch, one or two epochs on)
    # the schedule the Trainer stepped: warm-up over the first epoch's 2 steps, half cosine over the remaining ones (cosine.py:6-25)
    import math

    lrs = runs["auto"][1]
    want = [1e-3 * (s / 2 if s < 2 else 0.5 * (1 + math.cos(math.pi * (s - 2) / (n - 2)))) for s in range(1, n + 1)]
    assert lrs == pytest.approx(want, rel=1e-6), (lrs, want)
    AF.invalidate_weight_cache()


@pytest.mark.gpu
def test_trainer_native_step_replays_hipgraphs(monkeypatch):
    """On the MI355X the native mode's training_step is a graph replay from a shape's second visit on."""
    from auto_avsr_amd import _lib
    from auto_avsr_amd import functional as AF
    from auto_avsr_amd import train_native as TN

    _lib._lib = None
    dev = torch.device("cuda:0")
    mod, odim = _module(dev, "native", monkeypatch, "mixed")
    data = _Data(_batches(dev, odim, [(2, 12, 3), (3, 8, 4)]))
    seen = {}
    orig_close = TN.NativeStepper.close

    def close(self):
        seen.update(self.stats)
        orig_close(self)

    monkeypatch.setattr(TN.NativeStepper, "close", close)
    # (on the DEFAULT stream, as a Lightning Trainer would call it: the stepper moves the step onto a stream of its own)
    losses = StandInTrainer(data, max_epochs=4, gradient_clip_val=None).fit(mod)
    torch.cuda.synchronize()
    assert seen["captured"] == 2 and seen["replayed"] >= 6 and seen["eager"] == 2, seen
    assert all(l == l for l in losses) and losses[-1] < losses[0]
    AF.invalidate_weight_cache()
