"""Batch formation of the reference (datamodule/data_module.py:10-182): ``pad`` / ``collate_pad``, the length-bucketed,
frame-budgeted ``CustomBucketDataset`` and the ``DataModule`` that hands train / val / test loaders to the trainer.
With pytorch_lightning installed ``DataModule`` is a LightningDataModule (what train.py / eval.py pass to the Trainer);
without it the same class is a plain object with the same three loader methods (auto_avsr_amd.train_native / eval.py call
them directly).  File decoding lives in datamodule/av_dataset.py; the synthetic generator in auto_avsr_amd/synthetic.py."""
import os

import torch

from auto_avsr_amd.synthetic import bucket_batches

try:  # optional third-party harness (absent in the build image)
    from pytorch_lightning import LightningDataModule as _DMBase
except ImportError:  # pragma: no cover - exercised in this image
    _DMBase = object


def pad(samples, pad_val=0.0):
    """Right-pad a list of (T_i, ...) tensors to the longest; 1-D targets gain a channel dim (B, 1, L)."""
    lengths = [len(s) for s in samples]
    out = samples[0].new_full((len(samples), max(lengths)) + tuple(samples[0].shape[1:]), pad_val)
    for i, s in enumerate(samples):
        out[i, : len(s)] = s
    if samples[0].dim() == 1:
        out = out.unsqueeze(1)
    return out, lengths


def collate_pad(batch):
    res = {}
    for key in batch[0].keys():
        vals = [s[key] for s in batch if s[key] is not None]
        c, lens = pad(vals, -1 if key == "target" else 0.0)
        res[key + "s"] = c
        res[key + "_lengths"] = torch.tensor(lens)
    return res


class DeviceBatches:
    """Iterates a DataLoader of RAW items (AVDataset) and finishes every item in the MAIN process: upload, then the
    reference's per-sample transform chain + padding collation as one device launch per batch (transforms.video_batch /
    audio_batch).  `single` = the test loader (one utterance per item, no padding)."""

    def __init__(self, loader, dataset, device, single=False, epoch=0):
        self.loader, self.ds, self.device, self.single = loader, dataset, torch.device(device), single
        self.epoch = epoch  # first epoch this object serves (DataModule counts its train_dataloader() calls)
        self._seen = getattr(getattr(loader, "sampler", None), "epoch", None)

    def __len__(self):
        return len(self.loader)

    @property
    def sampler(self):  # (what a trainer looks at to find a DistributedSampler)
        return getattr(self.loader, "sampler", None)

    def _inputs(self, raws):
        from auto_avsr_amd import transforms as TR

        xs = [r.to(self.device, non_blocking=True) for r in raws]
        if self.ds.modality == "video":
            return TR.video_batch(xs, self.ds.video_transform.subset)
        at = self.ds.audio_transform
        return TR.audio_batch(xs, at.subset, at.add_noise)

    def __iter__(self):
        # This wrapper is not a DataLoader, so a trainer that injects a DistributedSampler into DataLoaders (Lightning) cannot
        # shard it: DataModule builds the sampler itself when a process group is up, and every pass over the loader is an epoch
        # (DistributedSampler.set_epoch: a fresh, rank-consistent shuffle per epoch).
        smp = self.sampler
        if hasattr(smp, "set_epoch"):
            # a trainer that found the sampler (Lightning, through `.sampler`) has already called set_epoch(current_epoch):
            # never override an epoch set from outside
            if getattr(smp, "epoch", None) == self._seen:  # (a custom sampler may have set_epoch() without an epoch attribute)
                smp.set_epoch(self.epoch)
            self._seen = getattr(smp, "epoch", None)
        self.epoch += 1
        for item in self.loader:
            if self.single:
                x, _ = self._inputs([item["input"]])
                yield {"input": x[0], "target": item["target"].to(self.device)}
                continue
            x, lens = self._inputs([s["input"] for s in item])
            t, tl = pad([s["target"].to(self.device) for s in item], -1)
            yield {"inputs": x, "input_lengths": torch.tensor(lens), "targets": t, "target_lengths": torch.tensor(tl)}


def _identity(x):
    return x


class CustomBucketDataset(torch.utils.data.Dataset):
    """Pre-formed batches: sort into `num_buckets` length buckets, pack greedily up to `max_frames` real frames."""

    def __init__(self, dataset, lengths, max_frames, num_buckets, shuffle=False, batch_size=None):
        super().__init__()
        assert len(dataset) == len(lengths)
        if shuffle or batch_size:
            raise NotImplementedError("shuffle / batch_size variants are unused by the reference's train.py")
        self.dataset = dataset
        self.batches = bucket_batches(lengths, max_frames, num_buckets)

    def __getitem__(self, idx):
        return [self.dataset[i] for i in self.batches[idx]]

    def __len__(self):
        return len(self.batches)


class DataModule(_DMBase):
    """data_module.py:109-182: train / val loaders yield pre-formed, padded batches (`batch_size=None`: one dataset item
    IS a batch), the test loader yields single utterances.  `args.synthetic_utterances = n` (an extra of this build)
    replaces the file-backed AVDataset by n synthetic utterances."""

    def __init__(self, args=None, batch_size=None, train_num_buckets=50, train_shuffle=True, num_workers=10, device=None):
        super().__init__()
        self.args = args
        self.batch_size = batch_size
        self.train_num_buckets = train_num_buckets
        self.train_shuffle = train_shuffle
        self.num_workers = num_workers
        # where the input transforms run (file-backed datasets): the GPU; the CPU only through the test suite's emulator build
        self.device = device or getattr(args, "device", None) or ("cuda" if torch.cuda.is_available() else "cpu")

    def _dataset(self, subset, label_file):
        from .av_dataset import AVDataset, SyntheticAVDataset
        from .transforms import AudioTransform, VideoTransform

        n_syn = getattr(self.args, "synthetic_utterances", 0)
        if n_syn:
            return SyntheticAVDataset(n_syn, self.args.modality, seed={"train": 0, "val": 1, "test": 2}[subset])
        # only the transform of the selected modality is built (the audio one loads the babble recording, which is data and
        # not part of this repository); at test time noise is added only for a real target SNR (999999 = the reference's "clean")
        at = vt = None
        if self.args.modality == "video":
            vt = VideoTransform(subset)
        elif subset == "test":
            snr = getattr(self.args, "decode_snr_target", 999999)
            at = AudioTransform("test", snr_target=snr if snr is not None and snr < 999999 else None)
        else:
            at = AudioTransform(subset)
        return AVDataset(root_dir=self.args.root_dir, label_path=os.path.join(self.args.root_dir, "labels", label_file),
                         subset=subset, modality=self.args.modality, audio_transform=at, video_transform=vt)

    def _finish(self, loader, dataset, single=False, epoch=0):
        return DeviceBatches(loader, dataset, self.device, single, epoch) if getattr(dataset, "raw", False) else loader

    def _train_epoch(self):
        """Epoch the next train loader starts at.  train.py reloads the loaders every epoch, so a per-loader counter would
        restart at 0 and every epoch would replay the same shuffle: the trainer's epoch when one is attached, else the number
        of loaders handed out so far (the same on every rank)."""
        tr = getattr(self, "trainer", None)
        ep = getattr(tr, "current_epoch", None)
        if ep is None:
            ep = self._train_loaders = getattr(self, "_train_loaders", -1) + 1
        return int(ep)

    def _workers(self):
        return 0 if getattr(self.args, "synthetic_utterances", 0) else self.num_workers

    @staticmethod
    def _dist_sampler(ds, shuffle):
        """One process per GPU (train.py:30-42): every rank must see its own share of the pre-formed batches.  The reference
        leaves that to Lightning, which swaps a DistributedSampler into plain DataLoaders; the loaders of this build may be
        wrapped (DeviceBatches), so the sampler is built here whenever a process group exists -- a trainer that finds a
        DistributedSampler already in place keeps it and only calls set_epoch()."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return None
        return torch.utils.data.distributed.DistributedSampler(ds, shuffle=shuffle, seed=0, drop_last=False)

    def train_dataloader(self):
        base = self._dataset("train", self.args.train_file)
        ds = CustomBucketDataset(base, base.input_lengths, self.args.max_frames, self.train_num_buckets,
                                 batch_size=self.batch_size)
        raw = getattr(base, "raw", False)
        smp = self._dist_sampler(ds, self.train_shuffle)
        return self._finish(torch.utils.data.DataLoader(ds, num_workers=self._workers(), batch_size=None,
                                                        shuffle=self.train_shuffle and smp is None, sampler=smp,
                                                        collate_fn=_identity if raw else collate_pad), base,
                            epoch=self._train_epoch())

    def val_dataloader(self):
        base = self._dataset("val", self.args.val_file)
        ds = CustomBucketDataset(base, base.input_lengths, 1000, 1, batch_size=self.batch_size)
        raw = getattr(base, "raw", False)
        return self._finish(torch.utils.data.DataLoader(ds, batch_size=None, num_workers=self._workers(), sampler=self._dist_sampler(ds, False),
                                                        collate_fn=_identity if raw else collate_pad), base)

    def test_dataloader(self):
        base = self._dataset("test", self.args.test_file)
        return self._finish(torch.utils.data.DataLoader(base, batch_size=None), base, single=True)
