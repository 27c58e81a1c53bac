"""Drop-in import path of the reference (espnet/nets/pytorch_backend/transformer/layer_norm.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import LayerNorm  # noqa: F401
