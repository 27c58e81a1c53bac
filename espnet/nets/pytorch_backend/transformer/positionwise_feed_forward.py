"""Drop-in import path of the reference (espnet/nets/pytorch_backend/transformer/positionwise_feed_forward.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import PositionwiseFeedForward  # noqa: F401
