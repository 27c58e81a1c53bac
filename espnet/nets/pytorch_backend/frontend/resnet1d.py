"""Drop-in import path of the reference (espnet/nets/pytorch_backend/frontend/resnet1d.py); implementation: auto_avsr_amd.frontend (HIP kernels)."""
from auto_avsr_amd.frontend import BasicBlock1D, Conv1dResNet, ResNet1D, audio_resnet  # noqa: F401
