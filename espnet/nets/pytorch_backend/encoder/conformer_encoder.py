"""Drop-in import path of the reference (espnet/nets/pytorch_backend/encoder/conformer_encoder.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import ConformerEncoder, ConvolutionModule, EncoderLayer  # noqa: F401
