"""Drop-in import path of the reference (espnet/nets/pytorch_backend/transformer/label_smoothing_loss.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import LabelSmoothingLoss  # noqa: F401
