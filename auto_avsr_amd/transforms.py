"""Input pipeline on the device: the reference's augmentation transforms and padding
collation behind the reference's own class names, executed by two batch-level kernels (csrc/augment.hip).

  reference (DataLoader workers, CPU, per sample)            here (one launch per batch, on the GPU that trains)
  -------------------------------------------------------   ---------------------------------------------------------
  VideoTransform(subset)            transforms.py:89-110     VideoTransform(subset)(clip)  /  video_batch(clips, subset)
  AudioTransform(subset, snr)       transforms.py:113-136    AudioTransform(subset, ...)(wav)  /  audio_batch(wavs, ...)
  AdaptiveTimeMask(window, stride)  transforms.py:44-64      AdaptiveTimeMask(window, stride).draw(length) -> intervals
  AddNoise(noise_filename, snr)     transforms.py:67-88      AddNoise(noise=tensor, snr_target).draw(length) -> (start, snr)
  pad / collate_pad                 data_module.py:10-41     fused into video_batch / audio_batch; pad_targets()

Every random decision (crop origin, masking runs, noise offset, SNR level) is drawn on the host with the SAME calls in the
SAME order as the reference classes make (torch.randint / random.randrange / random.randint / random.choice), so a seeded
run reproduces the reference's augmentation exactly; the kernels only apply the decisions.  Clips arrive in the video
decoder's layout (uint8 [T, H, W, 3]; the reference's load_video returns the [T, 3, H, W] permuted VIEW of exactly that)
and are uploaded as bytes: 3 B per pixel cross PCIe instead of the 4 B per pixel of a CPU-transformed f32 tensor.
There is no CPU implementation here: without libavsr_hip.so the calls raise (tests use the emulator build of the same
kernels)."""
import os
import random

import numpy as np

import torch

from . import ops

SNR_LEVELS = [-5, 0, 5, 10, 15, 20, 999999]  # transforms.py:75


class AdaptiveTimeMask:
    """transforms.py:44-64.  draw(length) consumes the RNGs exactly like the reference's forward and returns the
    [start, end) runs it would zero.  (Reference quirk, kept: of each random pair the first number only bounds the start
    position, the second is the run length.)"""

    def __init__(self, window, stride):
        self.window, self.stride = window, stride

    def draw(self, length):
        n_mask = int((length + self.stride - 0.1) // self.stride)
        ts = torch.randint(0, self.window, size=(n_mask, 2))
        out = []
        for t, width in ts.tolist():
            if length - t <= 0:
                continue
            t_start = random.randrange(0, length - t)
            if t == 0:
                continue
            out.append((t_start, min(t_start + width, length)))
        return out


class AddNoise:
    """transforms.py:67-88.  The recording is held on the device; draw(length) returns (start index, SNR in dB)."""

    def __init__(self, noise=None, snr_target=None, noise_filename=None):
        self.snr_levels = [snr_target] if snr_target else list(SNR_LEVELS)
        if noise is None:
            try:
                import torchaudio
            except ImportError as e:  # the image has no torchaudio: pass the recording as a tensor
                raise RuntimeError("AddNoise needs `noise` ([1, N] f32 tensor at 16 kHz) when torchaudio is absent") from e
            noise, rate = torchaudio.load(noise_filename)
            assert rate == 16000
        self.noise = noise.reshape(1, -1).to(torch.float32).contiguous()

    def draw(self, length):
        start = random.randint(0, self.noise.shape[1] - length)
        return start, random.choice(self.snr_levels)


def _table(device, ptrs, lens, extra_i32=(), intervals=None):
    """One host -> device copy carrying the per-sample tables of a launch; returns the device blob and the byte offsets
    (pointers first: 8-byte aligned)."""
    B = len(ptrs)
    max_iv = max([len(iv) for iv in intervals] + [1]) if intervals is not None else 0
    parts = [np.asarray(ptrs, dtype=np.int64).view(np.uint8)]
    offs = [0]
    for arr in (lens,) + tuple(extra_i32):
        offs.append(offs[-1] + parts[-1].size)
        parts.append(np.asarray(arr, dtype=np.int32).view(np.uint8))
    iv_off = niv_off = None
    if intervals is not None:
        niv = np.asarray([len(iv) for iv in intervals], dtype=np.int32)
        ivs = np.zeros((B, max_iv, 2), dtype=np.int32)
        for b, iv in enumerate(intervals):
            if iv:
                ivs[b, :len(iv)] = np.asarray(iv, dtype=np.int32)
        niv_off = offs[-1] + parts[-1].size
        parts.append(niv.view(np.uint8))
        iv_off = niv_off + niv.nbytes
        parts.append(ivs.reshape(-1).view(np.uint8))
    # blocking copy: the source is a temporary pageable buffer (an asynchronous copy could read it after it is gone)
    blob = torch.from_numpy(np.concatenate(parts)).to(device)
    return blob, offs, niv_off, iv_off, max_iv


def _as_thwc(clip):
    """uint8 [T, H, W, 3] contiguous view of a clip given as [T, H, W, 3] or as load_video's [T, 3, H, W] permutation."""
    assert clip.dtype == torch.uint8 and clip.dim() == 4, "clips are uint8 [T,H,W,3] or [T,3,H,W]"
    if clip.shape[-1] != 3 or (clip.shape[1] == 3 and not clip.is_contiguous()):
        clip = clip.permute(0, 2, 3, 1)  # TCHW -> THWC (free for load_video's permuted view)
    return clip.contiguous()


def video_batch(clips, subset, out_dtype=torch.float32, crop=88, mean=0.421, std=0.165):
    """VideoTransform(subset) on every clip + collate_pad, one launch.  clips: uint8 device tensors (decoder layout).
    Returns (batch [B, Tmax, 1, crop, crop], lengths list).  RNG order per clip as in the reference: crop, then mask."""
    clips = [_as_thwc(c) for c in clips]
    H, W = clips[0].shape[1:3]
    assert all(c.shape[1:3] == (H, W) for c in clips), "one frame size per batch"
    dev = clips[0].device
    lens = [c.shape[0] for c in clips]
    cy, cx, ivs = [], [], []
    masker = AdaptiveTimeMask(10, 25)
    for n in lens:
        if subset == "train":
            if (H, W) == (crop, crop):  # torchvision RandomCrop.get_params draws nothing when the sizes match
                i = j = 0
            else:
                i = torch.randint(0, H - crop + 1, size=(1,)).item()
                j = torch.randint(0, W - crop + 1, size=(1,)).item()
            ivs.append(masker.draw(n))
        else:
            i, j = int(round((H - crop) / 2.0)), int(round((W - crop) / 2.0))
            ivs.append([])
        cy.append(i)
        cx.append(j)
    B, Tmax = len(clips), max(lens)
    blob, offs, niv_off, iv_off, max_iv = _table(dev, [c.data_ptr() for c in clips], lens, (cy, cx), ivs)
    out = torch.empty(B, Tmax, 1, crop, crop, dtype=out_dtype, device=dev)
    base = blob.data_ptr()
    ops.call("avsr_video_transform", base + offs[0], base + offs[1], base + offs[2], base + offs[3], base + iv_off,
             base + niv_off, max_iv, ops._ptr(out), ops.dt(out), B, Tmax, H, W, crop, mean, std, ops._stream(out))
    _keep_alive(out, clips, blob)
    return out, lens


def audio_batch(wavs, subset, add_noise=None, eps=1e-8):
    """AudioTransform(subset) on every waveform ([T, 1] or [T] f32 device tensors) + collate_pad, one launch.
    add_noise: an AddNoise (always applied in training, as the reference does; in evaluation only with a target SNR).
    Returns (batch [B, Lmax, 1] f32, lengths list).  RNG order per utterance: mask, then noise offset, then SNR."""
    wavs = [w.reshape(-1).to(torch.float32).contiguous() for w in wavs]
    dev = wavs[0].device
    lens = [w.numel() for w in wavs]
    masker = AdaptiveTimeMask(6400, 16000)
    ivs, starts, snrs = [], [], []
    for n in lens:
        ivs.append(masker.draw(n) if subset == "train" else [])
        if add_noise is not None:
            s, snr = add_noise.draw(n)
            starts.append(s)
            snrs.append(float(snr))
        else:
            starts.append(-1)
            snrs.append(0.0)
    B, Lmax = len(wavs), max(lens)
    blob, offs, niv_off, iv_off, max_iv = _table(dev, [w.data_ptr() for w in wavs], lens, (), ivs)
    out = torch.empty(B, Lmax, 1, dtype=torch.float32, device=dev)
    base = blob.data_ptr()
    noise = start_t = snr_t = None
    if add_noise is not None:
        if add_noise.noise.device != dev:
            add_noise.noise = add_noise.noise.to(dev)
        noise = add_noise.noise
        start_t = torch.tensor(starts, dtype=torch.int64).to(dev)
        snr_t = torch.tensor(snrs, dtype=torch.float32).to(dev)
    ws = torch.empty(ops.call("avsr_audio_transform_workspace_bytes", B, Lmax), dtype=torch.uint8, device=dev)
    ops.call("avsr_audio_transform", base + offs[0], base + offs[1], base + iv_off, base + niv_off, max_iv, ops._ptr(noise),
             ops._ptr(start_t), ops._ptr(snr_t), eps, ops._ptr(out), B, Lmax, ops._ptr(ws), ops._stream(out))
    _keep_alive(out, wavs, blob, start_t, snr_t, ws)
    return out, lens


def _keep_alive(out, *objs):
    """The launch reads its inputs asynchronously: tie their lifetime to the output tensor."""
    out._avsr_inputs = objs


def pad_targets(targets, pad_val=-1):
    """collate_pad for the label sequences (data_module.py:10-41): [B, 1, Lmax] int64 padded with -1, plus lengths.
    Host-side bookkeeping on a few hundred integers."""
    lens = [len(t) for t in targets]
    out = torch.full((len(targets), 1, max(lens)), pad_val, dtype=torch.int64)
    for i, t in enumerate(targets):
        out[i, 0, :len(t)] = torch.as_tensor(t, dtype=torch.int64)
    return out, lens


class VideoTransform:
    """transforms.py:89-110 on one clip: [T, 3, H, W] uint8 (load_video) -> [T, 1, 88, 88] f32."""

    def __init__(self, subset):
        assert subset in ("train", "val", "test")
        self.subset = subset

    def __call__(self, sample):
        out, _ = video_batch([sample], self.subset)
        return out[0]


NOISE_FILENAME = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "datamodule", "babble_noise.wav")


def load_default_noise():
    """The reference's babble recording (datamodule/transforms.py:21-24: `babble_noise.wav` next to the module), read
    with the standard library: [1, n] f32 in [-1, 1).  The file is data, not code -- it is not part of this repository."""
    import wave

    import numpy as np

    if not os.path.exists(NOISE_FILENAME):
        raise FileNotFoundError(
            f"{NOISE_FILENAME} not found: AudioTransform needs the reference's babble noise recording for training-time / "
            "SNR-targeted augmentation -- copy it there, pass noise=<[1, n] tensor>, or pass noise=False to train without "
            "additive noise (NOT the reference's recipe)")
    with wave.open(NOISE_FILENAME, "rb") as f:
        assert f.getframerate() == 16000 and f.getsampwidth() == 2
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").reshape(-1, f.getnchannels())
    return torch.from_numpy(pcm[:, 0].astype(np.float32) / 32768.0).unsqueeze(0)


class AudioTransform:
    """transforms.py:113-136 on one waveform: [T, 1] f32 (load_audio) -> [T, 1] f32.
    noise: [1, n] babble recording; None = load the reference's default file (raises if it is absent -- the reference
    always adds noise in training, transforms.py:116-118); False = explicitly no additive noise."""

    def __init__(self, subset, snr_target=None, noise=None):
        assert subset in ("train", "val", "test")
        self.subset = subset
        self.add_noise = None
        if noise is not False and (subset == "train" or snr_target is not None):
            if noise is None:
                noise = load_default_noise()
            self.add_noise = AddNoise(noise=noise, snr_target=None if subset == "train" else snr_target)

    def __call__(self, sample):
        out, _ = audio_batch([sample], self.subset, self.add_noise)
        return out[0]
