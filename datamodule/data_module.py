"""Batch formation of the reference (datamodule/data_module.py:10-106) without its file I/O: ``pad`` /
``collate_pad`` and the length-bucketed, frame-budgeted batching.  The LightningDataModule wrapper and the
mp4 / wav readers need pytorch_lightning, torchvision and torchaudio and are out of the hot-path scope; the
synthetic generator used for measurement lives in auto_avsr_amd/synthetic.py."""
import torch

from auto_avsr_amd.synthetic import bucket_batches


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
