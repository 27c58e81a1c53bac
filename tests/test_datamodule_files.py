"""File-backed data path (datamodule/av_dataset.py AVDataset behind DataModule's three loaders): decoding on the host, the
reference's transforms + collation on the device, in the main process.  Audio reads real wav files written by the test; video
decoding needs torchvision (absent), so `load_video` is replaced by a synthetic decoder -- everything downstream is the product."""
import os
import types
import wave

import numpy as np
import pytest
import torch

from auto_avsr_amd import transforms as TR


def _write_tree(root, n, modality):
    os.makedirs(os.path.join(root, "labels"), exist_ok=True)
    os.makedirs(os.path.join(root, "lrs3", "clips"), exist_ok=True)
    rng = np.random.default_rng(0)
    rows, wavs = [], {}
    for i in range(n):
        frames = int(rng.integers(5, 14))
        rel = f"clips/u{i}.mp4"
        pcm = (rng.standard_normal(frames * 640) * 3000).astype("<i2")
        with wave.open(os.path.join(root, "lrs3", rel[:-4] + ".wav"), "wb") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(16000)
            f.writeframes(pcm.tobytes())
        wavs[rel] = pcm
        ids = " ".join(str(int(v)) for v in rng.integers(1, 30, size=max(1, frames // 3)))
        rows.append(f"lrs3,{rel},{frames},{ids}")
    for name in ("train.csv", "val.csv", "test.csv"):
        with open(os.path.join(root, "labels", name), "w") as f:
            f.write("\n".join(rows))
    return rows, wavs


@pytest.mark.parametrize("modality", ["audio", "video"])
def test_datamodule_file_backed_loaders(dev, tmp_path, monkeypatch, modality):
    from datamodule import av_dataset
    from datamodule.data_module import DataModule

    root = str(tmp_path)
    rows, wavs = _write_tree(root, 7, modality)
    clips = {}

    def fake_load_video(path):  # [T, 3, 96, 96] uint8, as torchvision.io.read_video(...).permute(0, 3, 1, 2) would give
        rel = os.path.relpath(path, os.path.join(root, "lrs3"))
        t = int([r for r in rows if rel in r][0].split(",")[2])
        g = torch.Generator().manual_seed(len(rel) + t)
        clips[rel] = torch.randint(0, 256, (t, 96, 96, 3), generator=g, dtype=torch.uint8)
        return clips[rel].permute(0, 3, 1, 2)

    monkeypatch.setattr(av_dataset, "load_video", fake_load_video)
    args = types.SimpleNamespace(root_dir=root, modality=modality, train_file="train.csv", val_file="val.csv", test_file="test.csv",
                                 max_frames=30, synthetic_utterances=0)
    dm = DataModule(args, num_workers=0, device=str(dev))
    if modality == "audio":
        monkeypatch.setattr(TR, "load_default_noise", lambda: torch.randn(1, 40000, generator=torch.Generator().manual_seed(3)))
    seen = 0
    for b in dm.train_dataloader():
        B = b["inputs"].shape[0]
        assert b["inputs"].device.type == dev.type and b["targets"].shape[:2] == (B, 1)
        per = 640 if modality == "audio" else 1
        assert b["inputs"].shape[1] == int(b["input_lengths"].max()) and int(b["input_lengths"].sum()) // per <= 30
        assert b["inputs"].shape[2:] == ((1,) if modality == "audio" else (1, 88, 88))
        assert torch.isfinite(b["inputs"].float()).all()
        seen += B
    assert seen == 7
    val = list(dm.val_dataloader())
    assert sum(v["inputs"].shape[0] for v in val) == 7
    tests = list(dm.test_dataloader())
    assert len(tests) == 7
    # the evaluation transforms are deterministic: the loader's item equals the transform applied by hand
    rel = rows[0].split(",")[1]
    if modality == "audio":
        wav = torch.from_numpy(wavs[rel].astype(np.float32) / 32768.0).view(-1, 1)
        want = TR.AudioTransform("test")(wav.to(dev))
    else:
        want = TR.VideoTransform("test")(clips[rel].permute(0, 3, 1, 2).to(dev))
    assert torch.equal(tests[0]["input"].cpu(), want.cpu())
    assert tests[0]["target"].tolist() == [int(v) for v in rows[0].split(",")[3].split()]


def test_datamodule_shards_wrapped_loaders_across_ranks(dev, tmp_path, monkeypatch):
    """Round-3 advisor finding: the file-backed loaders are DeviceBatches wrappers, which a trainer cannot give a
    DistributedSampler -- every rank would train on every batch.  With a process group up the DataModule shards them itself:
    the ranks' batches are disjoint, cover the epoch, and the shuffle changes per epoch (set_epoch) consistently across ranks."""
    import torch.distributed as dist

    from datamodule.data_module import DataModule

    root = str(tmp_path)
    _write_tree(root, 23, "audio")
    monkeypatch.setattr(TR, "load_default_noise", lambda: torch.randn(1, 40000, generator=torch.Generator().manual_seed(3)))
    args = types.SimpleNamespace(root_dir=root, modality="audio", train_file="train.csv", val_file="val.csv", test_file="test.csv",
                                 max_frames=30, synthetic_utterances=0)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    seen = {}
    for rank in (0, 1):
        monkeypatch.setattr(dist, "get_rank", lambda group=None, r=rank: r)
        dm = DataModule(args, num_workers=0, device=str(dev))
        loader = dm.train_dataloader()
        assert isinstance(loader.sampler, torch.utils.data.distributed.DistributedSampler)
        epochs = []
        for _ in range(2):  # two passes = two epochs
            epochs.append([tuple(b["input_lengths"].tolist()) + tuple(b["targets"].flatten().tolist()) for b in loader])
        seen[rank] = epochs
        val = dm.val_dataloader()
        assert isinstance(val.sampler, torch.utils.data.distributed.DistributedSampler) and not val.sampler.shuffle
    nb = len(dm.train_dataloader().loader.dataset)
    for e in (0, 1):
        a, b = seen[0][e], seen[1][e]
        assert len(a) == len(b) == (nb + 1) // 2
        assert len(set(a) | set(b)) == nb          # the two ranks cover every batch of the epoch ...
        assert len(set(a) & set(b)) == 2 * len(a) - nb  # ... sharing only the DistributedSampler's padding batch (odd counts)
    assert seen[0][0] != seen[0][1]                # a fresh shuffle per epoch


def test_reloaded_train_loaders_reshuffle_and_schedule_length(dev, tmp_path, monkeypatch):
    """Round-4 advisor findings.  (1) train.py reloads the loaders every epoch: each fresh DeviceBatches must start at the
    DataModule's epoch, not at 0 (the same batch order every epoch), and must not override an epoch a trainer set through
    `.sampler`.  (2) a loader that carries its own DistributedSampler already reports the PER-RANK batch count:
    lightning.steps_per_epoch must not divide it by the world size again (the cosine schedule would end after 1 / world)."""
    import torch.distributed as dist

    import lightning as LM
    from datamodule.data_module import DataModule

    root = str(tmp_path)
    _write_tree(root, 23, "audio")
    monkeypatch.setattr(TR, "load_default_noise", lambda: torch.randn(1, 40000, generator=torch.Generator().manual_seed(3)))
    args = types.SimpleNamespace(root_dir=root, modality="audio", train_file="train.csv", val_file="val.csv", test_file="test.csv",
                                 max_frames=30, synthetic_utterances=0)
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "get_rank", lambda group=None: 0)
    dm = DataModule(args, num_workers=0, device=str(dev))
    key = lambda loader: [tuple(b["input_lengths"].tolist()) + tuple(b["targets"].flatten().tolist()) for b in loader]  # noqa: E731
    first, second = key(dm.train_dataloader()), key(dm.train_dataloader())  # one fresh loader per epoch, as train.py asks for
    assert first != second and sorted(first) != [] and len(first) == len(second)
    # a trainer's set_epoch wins: two loaders given the same epoch from outside replay the same order
    la, lb = dm.train_dataloader(), dm.train_dataloader()
    la.sampler.set_epoch(7)
    lb.sampler.set_epoch(7)
    ka = key(la)
    assert ka == key(lb)
    # with a trainer attached the epoch comes from it
    dm.trainer = types.SimpleNamespace(current_epoch=7)
    assert key(dm.train_dataloader()) == ka
    del dm.trainer
    # schedule length: per-rank count taken as it is when the loader is sharded, divided when it is not
    loader = dm.train_dataloader()
    nb = len(loader.loader.dataset)
    assert len(loader) == (nb + 1) // 2
    assert LM.steps_per_epoch(loader, 2) == (nb + 1) // 2
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    plain = DataModule(args, num_workers=0, device=str(dev)).train_dataloader()
    assert len(plain) == nb and LM.steps_per_epoch(plain, 2) == nb / 2


def test_native_fit_on_file_backed_data(dev, tmp_path, monkeypatch):
    """train.py's native loop over FILES: `train_native.fit` with `--train-file` set draws its batches from
    DataModule.train_dataloader() -- AVDataset reading wav files, CustomBucketDataset's length-bucketed batches, the reference's
    training transform (time masking + babble noise) and padding collation as one device launch per batch -- instead of the
    synthetic corpus; a validation pass over val_dataloader() closes each epoch."""
    from auto_avsr_amd import functional as AF
    from auto_avsr_amd import train_native as TN
    from auto_avsr_amd.e2e import E2E

    root = str(tmp_path)
    _write_tree(root, 9, "audio")
    monkeypatch.setattr(TR, "load_default_noise", lambda: torch.randn(1, 40000, generator=torch.Generator().manual_seed(3)))
    torch.manual_seed(0)
    AF.invalidate_weight_cache()
    m = E2E(31, "audio", adim=128, aheads=2, eunits=64, elayers=1, dunits=64, dlayers=1, cnn_module_kernel=7).to(dev).train()
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    args = types.SimpleNamespace(root_dir=root, modality="audio", train_file="train.csv", val_file="val.csv", test_file="test.csv",
                                 max_frames=30, train_num_buckets=3, lr=1e-3, weight_decay=0.03, warmup_epochs=0, max_epochs=2,
                                 exp_dir=None, exp_name="run", ckpt_path=None, steps=None, val_batches=1, synthetic=False,
                                 synthetic_utterances=0, log_every=1, numerics="precise" if dev.type == "cpu" else "mixed", num_workers=0)
    logs = []
    losses = TN.fit(m, args, dev, log=logs.append)
    assert isinstance(TN._batch_source(args, m, dev, 0, 1), TN._FileSource)
    assert len(losses) >= 4 and all(v == v and abs(v) < 1e6 for v in losses)
    assert sum("validation" in s for s in logs) == 2
    moved = sum(float((v.float() - before[k].float()).abs().sum()) for k, v in m.state_dict().items() if v.is_floating_point())
    assert moved > 0 and all(torch.isfinite(v).all() for v in m.state_dict().values() if v.is_floating_point())
    AF.invalidate_weight_cache()
