"""Known-answer vectors for the five THIRD-PARTY operations on the reference's input path whose libraries (torchvision,
torchaudio) exist neither in /root/reference nor in this image: RandomCrop / CenterCrop / Grayscale / Normalize
(datamodule/transforms.py:92-105) and torchaudio.functional.add_noise (:85).

The expected values are NOT produced by oracle/transforms_oracle.py or by the kernels: they are computed here from the
libraries' PUBLISHED definitions with exact rational arithmetic (fractions.Fraction) and rounded once to float32 --
  Grayscale        l = 0.2989 r + 0.587 g + 0.114 b                       (torchvision rgb_to_grayscale, ITU-R 601-2 luma)
  Normalize        (x - mean) / std                                       (torchvision functional.normalize)
  CenterCrop       top = round((H - h) / 2), left = round((W - w) / 2)    (torchvision functional.center_crop)
  RandomCrop       i = randint(0, H - h + 1), then j = randint(0, W - w + 1), both torch.randint(size=(1,)) draws
                   (torchvision RandomCrop.get_params); the vectors record torch's own draws for fixed seeds
  add_noise        y = x + sqrt(E_x / E_n * 10^(-snr / 10)) * n,  E = sum of squares over the last axis
                   (torchaudio.functional.add_noise without `lengths`)
so a wrong restatement on either side (oracle or kernels) shows up as a disagreement with arithmetic done a third way.
    python tests/golden/make_golden_thirdparty.py   ->  tests/golden/golden_thirdparty_v1.json"""
import json
import math
import os
import struct
from fractions import Fraction as F

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def f32(x):
    """nearest float32 of a Fraction / float, as a python float"""
    return struct.unpack("f", struct.pack("f", float(x)))[0]


def main():
    out = {}
    # ---- Grayscale + Normalize on exact 8-bit colours (the pipeline divides by 255 first: transforms.py:91)
    rgb = [(0, 0, 0), (255, 255, 255), (100, 150, 200), (17, 231, 64), (255, 0, 0), (0, 255, 0), (0, 0, 255), (128, 128, 127)]
    gray, norm = [], []
    for r, g, b in rgb:
        l = F(2989, 10000) * F(r, 255) + F(587, 1000) * F(g, 255) + F(114, 1000) * F(b, 255)
        gray.append(f32(l))
        norm.append(f32((l - F(421, 1000)) / F(165, 1000)))
    out["rgb"] = rgb
    out["gray"] = gray
    out["gray_normalized"] = norm
    # ---- CenterCrop offsets
    out["center_crop"] = [{"H": H, "W": W, "size": s, "top": int(round((H - s) / 2.0)), "left": int(round((W - s) / 2.0))}
                          for (H, W, s) in [(96, 96, 88), (97, 96, 88), (88, 88, 88), (120, 101, 88)]]
    # ---- RandomCrop.get_params: two torch.randint(0, n, size=(1,)) draws, row offset first
    rc = []
    for seed in (0, 1, 7):
        torch.manual_seed(seed)
        i = int(torch.randint(0, 96 - 88 + 1, size=(1,)))
        j = int(torch.randint(0, 96 - 88 + 1, size=(1,)))
        rc.append({"seed": seed, "H": 96, "W": 96, "size": 88, "i": i, "j": j})
    out["random_crop"] = rc
    # ---- add_noise: small integer signals, exact energies
    cases = []
    for x, n, snr in [([3, 4, 0, 0], [1, 0, 0, 0], 0), ([3, 4, 0, 0], [1, 0, 0, 0], 20), ([1, -2, 2, 4], [2, 2, -2, 2], 5),
                      ([6, 0, 8, 0, 0], [0, 3, 0, 4, 0], -5)]:
        ex, en = sum(v * v for v in x), sum(v * v for v in n)
        scale = math.sqrt(ex / en * 10.0 ** (-snr / 10.0))
        cases.append({"x": x, "n": n, "snr_db": snr, "y": [f32(a + scale * b) for a, b in zip(x, n)]})
    out["add_noise"] = cases
    with open(os.path.join(HERE, "golden_thirdparty_v1.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote golden_thirdparty_v1.json")


if __name__ == "__main__":
    main()
