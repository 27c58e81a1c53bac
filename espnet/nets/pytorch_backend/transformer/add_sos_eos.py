"""Drop-in import path of the reference (espnet/nets/pytorch_backend/transformer/add_sos_eos.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import add_sos_eos  # noqa: F401
