"""Deterministic synthetic weights / inputs shared by the golden-vector generator and the tests.

Weights are a pure function of (parameter name, shape, seed), so the reference model (in the build container)
and the implementation under test (anywhere) can be given bit-identical parameters without shipping them.
Scales are chosen non-degenerate (default-init eval-mode outputs are ~1e-4 and useless for
relative-error tests): unit-gain weights, randomised BatchNorm running statistics, noisy norm gains."""
import zlib

import torch


def _gen(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synth_tensor(name, shape, dtype, seed=0):
    g = _gen(name, seed)
    shape = tuple(shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=dtype)
    if name.endswith("running_var"):
        return torch.rand(shape, generator=g) + 0.5
    if name.endswith("running_mean"):
        return torch.randn(shape, generator=g) * 0.1
    if "pos_bias_" in name:
        return torch.randn(shape, generator=g) * 0.3
    if len(shape) == 1:
        if name.endswith("weight"):  # norm gains
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)  # biases
    if "embed" in name:
        return torch.randn(shape, generator=g) * 0.05
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) / fan_in ** 0.5


def synth_state_dict(template, seed=0):
    """template: a state_dict (or {name: tensor}) giving names, shapes and dtypes."""
    return {k: synth_tensor(k, v.shape, v.dtype, seed) for k, v in template.items()}


def synth_batch(modality, B, T, L, odim, seed=0, lengths=None):
    g = torch.Generator().manual_seed(1234 + seed)
    if lengths is None:
        lengths = [T - 3 * i for i in range(B)]
    lengths = torch.tensor(lengths, dtype=torch.int64)
    if modality == "video":
        x = torch.randn(B, T, 1, 88, 88, generator=g)
        for b in range(B):
            x[b, lengths[b]:] = 0
    else:
        x = torch.randn(B, T * 640, 1, generator=g)
        for b in range(B):
            x[b, lengths[b] * 640:] = 0
        lengths = lengths * 640
    y = torch.randint(1, odim - 1, (B, 1, L), generator=g)
    for b in range(1, B):
        y[b, 0, L - b:] = -1
    return x, lengths, y
