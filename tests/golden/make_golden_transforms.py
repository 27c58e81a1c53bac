"""Generates tests/golden/golden_transforms_v1.pt by running the REFERENCE's own datamodule/transforms.py
(VideoTransform / AudioTransform / AdaptiveTimeMask / AddNoise) and datamodule/data_module.py (pad) in this container.

The reference file imports torchvision and torchaudio, which are not installed here.  They are replaced by stub modules
that carry the restated published algorithms of the five library calls the file makes (oracle/transforms_oracle.py --
parity unpinned for those five); everything the reference itself defines (AdaptiveTimeMask's RNG protocol and quirks,
AddNoise's segment / SNR draws, the composition order of both pipelines, pad) runs from the reference source.

Run:  python tests/golden/make_golden_transforms.py     (needs /root/reference; the committed .pt does not)
"""
import importlib.util
import os
import random
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import transforms_oracle as TO  # noqa: E402

REF = "/root/reference"
NOISE = None  # set below; what the stubbed torchaudio.load returns


def install_stubs():
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")

    class RandomCrop(torch.nn.Module):
        def __init__(self, size):
            super().__init__()
            self.size = size

        def forward(self, img):
            i, j = TO.random_crop_params(img.shape[-2], img.shape[-1], self.size)
            return img[..., i:i + self.size, j:j + self.size]

    class CenterCrop(torch.nn.Module):
        def __init__(self, size):
            super().__init__()
            self.size = size

        def forward(self, img):
            i, j = TO.center_crop_params(img.shape[-2], img.shape[-1], self.size)
            return img[..., i:i + self.size, j:j + self.size]

    class Grayscale(torch.nn.Module):
        def forward(self, img):
            return TO.rgb_to_grayscale(img)

    class Normalize(torch.nn.Module):
        def __init__(self, mean, std):
            super().__init__()
            self.mean, self.std = mean, std

        def forward(self, t):
            return TO.normalize(t, self.mean, self.std)

    tv.transforms.RandomCrop, tv.transforms.CenterCrop = RandomCrop, CenterCrop
    tv.transforms.Grayscale, tv.transforms.Normalize = Grayscale, Normalize
    ta = types.ModuleType("torchaudio")
    ta.functional = types.ModuleType("torchaudio.functional")
    ta.functional.add_noise = TO.add_noise
    ta.load = lambda path, **kw: (NOISE, 16000)
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningDataModule = object
    for name, mod in (("torchvision", tv), ("torchvision.transforms", tv.transforms), ("torchaudio", ta),
                      ("torchaudio.functional", ta.functional), ("pytorch_lightning", pl)):
        sys.modules[name] = mod


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    global NOISE
    install_stubs()
    g = torch.Generator().manual_seed(2024)
    NOISE = torch.randn(1, 16000 * 6, generator=g) * 0.05
    T = load(os.path.join(REF, "datamodule", "transforms.py"), "ref_transforms")
    cases = {"noise_recipe": "torch.randn(1, 96000, generator=torch.Generator().manual_seed(2024)) * 0.05",
             "noise_probe": NOISE[0, ::9600].clone()}
    # ---- video: train (random crop + time mask) and val (centre crop), clip [T, 3, 96, 96] uint8 as load_video gives it
    for tag, subset, frames, seed in (("video_train", "train", 61, 7), ("video_train2", "train", 26, 8), ("video_val", "val", 30, 9)):
        clip = torch.randint(0, 256, (frames, 96, 96, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(seed))
        torch.manual_seed(seed)
        random.seed(seed)
        out = T.VideoTransform(subset)(clip.permute(0, 3, 1, 2))
        cases[tag] = dict(subset=subset, frames=frames, seed=seed, sample_frames=out[::12].clone(),
                          sum=float(out.double().sum()), abssum=float(out.double().abs().sum()), shape=tuple(out.shape))
    # ---- AdaptiveTimeMask alone (its RNG protocol), on index ramps so the zeroed runs can be read back
    masks = []
    for length, window, stride, seed in ((61, 10, 25, 1), (400, 10, 25, 2), (12, 10, 25, 3), (48000, 6400, 16000, 4), (9, 10, 25, 5)):
        torch.manual_seed(seed)
        random.seed(seed)
        y = T.AdaptiveTimeMask(window, stride)(torch.arange(1, length + 1, dtype=torch.float32))
        masks.append(dict(length=length, window=window, stride=stride, seed=seed, zero=(y == 0).nonzero().flatten().clone()))
    cases["masks"] = masks
    # ---- audio: train (mask + noise at a drawn SNR + layer_norm), val clean, val at a target SNR
    for tag, subset, snr_target, n, seed in (("audio_train", "train", None, 40000, 11), ("audio_train2", "train", None, 23000, 12),
                                             ("audio_val", "val", None, 16000, 13), ("audio_val_snr", "val", 5, 20000, 14)):
        wav = torch.randn(n, 1, generator=torch.Generator().manual_seed(seed)) * 0.1
        torch.manual_seed(seed)
        random.seed(seed)
        out = T.AudioTransform(subset, snr_target=snr_target)(wav)
        cases[tag] = dict(subset=subset, snr_target=snr_target, n=n, seed=seed, out_every4=out[::4].clone(),
                          sum=float(out.double().sum()), sqsum=float((out.double() ** 2).sum()))
    # ---- pad (data_module.py:10-41)
    samples = [torch.arange(5.0).view(5, 1), torch.arange(3.0).view(3, 1), torch.arange(4.0).view(4, 1)]
    src = open(os.path.join(REF, "datamodule", "data_module.py")).read()
    ns = {"torch": torch}
    start, end = src.index("def pad("), src.index("def collate_pad(")
    exec(compile(src[start:end], "ref_pad", "exec"), ns)  # the reference's own `pad` (the module has package-relative imports)
    batch, lengths = ns["pad"](samples, 0.0)
    tb, tl = ns["pad"]([torch.tensor([3, 4, 5]), torch.tensor([7])], -1)
    cases["pad"] = dict(batch=batch, lengths=lengths, target_batch=tb, target_lengths=tl)
    torch.save(cases, os.path.join(HERE, "golden_transforms_v1.pt"))
    print("wrote golden_transforms_v1.pt:", {k: (v.get("shape") if isinstance(v, dict) else None) for k, v in cases.items()})


if __name__ == "__main__":
    main()
