"""Drop-in import path of the reference (espnet/nets/pytorch_backend/transformer/mask.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import subsequent_mask, target_mask  # noqa: F401
