"""espnet.nets.e2e_asr_common (reference import path): end detection of the beam search."""
from auto_avsr_amd.decoding import end_detect  # noqa: F401
