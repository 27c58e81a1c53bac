"""Utterance datasets behind the reference's `AVDataset` contract (datamodule/av_dataset.py:29-73): item =
{"input": transformed clip, "target": token ids}; `input_lengths` (frames) drives the length bucketing.

* `AVDataset` reads the reference's on-disk layout -- `<root>/<dataset>/<rel_path>` mp4 (96x96 RGB, 25 fps) + the wav next
  to it, listed by a CSV of `dataset,rel_path,frames,token ids`.  Decoding needs torchvision (mp4) -- absent from this
  image, so video raises a clear ImportError at read time; wav files are read with the standard library when torchaudio
  is missing.  File I/O stays on the host.
* `SyntheticAVDataset` produces LRS3-shaped utterances with no files: what `eval.py` / `train.py
  --synthetic` iterate in this image."""
import os

import torch


def load_video(path):
    """T x C x H x W uint8 frames (av_dataset.py:13-19)."""
    try:
        import torchvision
    except ImportError as e:  # pragma: no cover - image without torchvision
        raise ImportError("reading mp4 clips needs torchvision.io (not installed in this image)") from e
    vid = torchvision.io.read_video(path, pts_unit="sec", output_format="THWC")[0]
    return vid.permute((0, 3, 1, 2))


def load_audio(path):
    """T x 1 float waveform in [-1, 1) from the wav next to the clip (av_dataset.py:22-27)."""
    wav = path[:-4] + ".wav"
    try:
        import torchaudio

        waveform, _ = torchaudio.load(wav, normalize=True)
        return waveform.transpose(1, 0)
    except ImportError:
        import wave

        import numpy as np

        with wave.open(wav, "rb") as f:
            assert f.getsampwidth() == 2, "16-bit PCM expected"
            pcm = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").reshape(-1, f.getnchannels())
        return torch.from_numpy(pcm.astype(np.float32) / 32768.0)[:, :1]


class AVDataset(torch.utils.data.Dataset):
    """Items are RAW: {"input": decoded clip [T, 3, H, W] uint8 / waveform [T, 1] f32 on the HOST, "target": token ids}.
    The reference applies `video_transform` / `audio_transform` here, on the CPU, inside DataLoader workers; this build's
    transforms are device kernels (auto_avsr_amd/transforms.py, csrc/augment.hip), which can run neither on host tensors nor in
    forked workers -- so decoding stays in `__getitem__` (workers welcome) and `DataModule` applies the transform + padding
    collation as ONE launch per batch in the main process, after the upload (`raw = True` tells it to).  The two transform
    objects are kept for the reference's constructor signature and carry the configuration (subset, noise, target SNR)."""

    raw = True

    def __init__(self, root_dir, label_path, subset, modality, audio_transform, video_transform, rate_ratio=640):
        self.root_dir, self.modality, self.rate_ratio, self.subset = root_dir, modality, rate_ratio, subset
        self.list = self.load_list(label_path)
        self.input_lengths = [int(row[2]) for row in self.list]
        self.audio_transform, self.video_transform = audio_transform, video_transform

    @staticmethod
    def load_list(label_path):
        rows = []
        with open(label_path) as f:
            for line in f.read().splitlines():
                dataset_name, rel_path, input_length, token_id = line.split(",")
                rows.append((dataset_name, rel_path, int(input_length), torch.tensor([int(t) for t in token_id.split()])))
        return rows

    def __getitem__(self, idx):
        dataset_name, rel_path, _, token_id = self.list[idx]
        path = os.path.join(self.root_dir, dataset_name, rel_path)
        if self.modality == "video":
            return {"input": load_video(path), "target": token_id}
        return {"input": load_audio(path), "target": token_id}

    def __len__(self):
        return len(self.list)


class SyntheticAVDataset(torch.utils.data.Dataset):
    """`n` synthetic utterances in the form the model consumes AFTER the reference's transforms: video (T, 1, 88, 88)
    z-normalised noise, audio (640 T, 1) layer-normed noise; targets round(T / 6.5) ids uniform in [1, odim - 2]."""

    def __init__(self, n, modality="video", odim=5049, seed=0, lengths=None, device="cpu"):
        from auto_avsr_amd.synthetic import utterance_lengths

        self.modality, self.odim, self.seed, self.device = modality, odim, seed, device
        self.input_lengths = [int(v) for v in (lengths if lengths is not None else utterance_lengths(n, seed=42 + seed))]

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(7919 * self.seed + idx)
        t = self.input_lengths[idx]
        target = torch.randint(1, self.odim - 1, (max(1, int(round(t / 6.5))),), generator=g)
        if self.modality == "video":
            x = torch.randn(t, 1, 88, 88, generator=g)
        else:
            w = torch.randn(t * 640, generator=g)
            x = ((w - w.mean()) / w.std()).unsqueeze(1)
        return {"input": x.to(self.device), "target": target.to(self.device)}

    def __len__(self):
        return len(self.input_lengths)
