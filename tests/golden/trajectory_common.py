"""Shared by make_golden_trajectory.py (runs the reference) and tests/test_trajectory.py: model sizes, schedule, the four batches."""
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden_trajectory_v1.pt")
ODIM, D, H, U, NENC, NDEC, SEED = 60, 128, 2, 256, 2, 2, 41
WARMUP, TOTAL = 5, 400  # WarmupCosineScheduler(warmup_epochs=5, total_epochs=400, steps_per_epoch=1): per-step schedule
SHAPES = [(2, 14, 3), (3, 10, 2), (2, 18, 4), (4, 8, 2)]  # (B, T, L)


def batch(k):
    """Batch k of the cycle: x (B, T, 1, 88, 88) zero-padded, lengths, labels (B, 1, L) padded with -1."""
    B, T, L = SHAPES[k]
    g = torch.Generator().manual_seed(8100 + k)
    lens = torch.tensor([T - 2 * (b % 2) for b in range(B)], dtype=torch.int64)
    x = torch.randn(B, T, 1, 88, 88, generator=g)
    y = torch.randint(1, ODIM - 1, (B, 1, L), generator=g)
    for b in range(B):
        x[b, lens[b]:] = 0
        if b % 2:
            y[b, 0, L - 1:] = -1
    return x, lens, y
