"""Shared by make_golden_trained.py (runs the reference) and tests/test_wer_trained.py: the small model's sizes, the 32 synthetic
utterances (lengths, videos, label strings -- all regenerated from seeds) and the word-level edit distance.  No reference import
here: this module travels to the GPU box."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden_trained_v1.pt")
ODIM, D, H, U, NENC, NDEC, SEED, BEAM, NUTT = 200, 128, 2, 512, 3, 2, 31, 40, 32


def lengths():
    """32 utterance lengths spread geometrically over the reference's range of 12 ... 400 frames (0.5 - 16 s at 25 fps)."""
    return [int(v) for v in np.round(np.geomspace(12, 400, NUTT))]


def video(i, T):
    g = torch.Generator().manual_seed(7000 + i)
    return torch.randn(T, 1, 88, 88, generator=g)


def labels():
    """Label strings: max(1, round(T / 8)) token ids in [1, ODIM - 2] per utterance."""
    g = torch.Generator().manual_seed(7777)
    return [torch.randint(1, ODIM - 1, (max(1, round(T / 8)),), generator=g).tolist() for T in lengths()]


def edit_distance(ref, hyp):
    """Word-level Levenshtein distance (lightning.py:12-14 compute_word_level_distance on token-id 'words')."""
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]
