"""Drop-in import path of the reference (espnet/nets/pytorch_backend/decoder/transformer_decoder.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import DecoderLayer, TransformerDecoder  # noqa: F401
