"""Drop-in import path of the reference (espnet/nets/pytorch_backend/frontend/resnet.py); implementation: auto_avsr_amd.frontend (HIP kernels)."""
from auto_avsr_amd.frontend import BasicBlock, Conv3dResNet, ResNet, conv3x3, downsample_basic_block, threeD_to_2D_tensor, video_resnet  # noqa: F401
