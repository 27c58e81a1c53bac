"""Drop-in import path of the reference (espnet/nets/pytorch_backend/e2e_asr_conformer.py); implementation: auto_avsr_amd.e2e (HIP kernels)."""
from auto_avsr_amd.e2e import E2E  # noqa: F401
