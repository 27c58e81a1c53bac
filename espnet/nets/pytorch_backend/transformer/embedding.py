"""Drop-in import path of the reference (espnet/nets/pytorch_backend/transformer/embedding.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import PositionalEncoding, RelPositionalEncoding, ScaledPositionalEncoding  # noqa: F401
