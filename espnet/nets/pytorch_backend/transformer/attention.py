"""Drop-in import path of the reference (espnet/nets/pytorch_backend/transformer/attention.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import MultiHeadedAttention, RelPositionMultiHeadedAttention  # noqa: F401
