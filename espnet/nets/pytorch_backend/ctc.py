"""Drop-in import path of the reference (espnet/nets/pytorch_backend/ctc.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import CTC  # noqa: F401
