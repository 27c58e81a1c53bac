"""espnet.nets.batch_beam_search (reference import path) -> auto_avsr_amd.decoding."""
from auto_avsr_amd.decoding import BatchBeamSearch, Hypothesis  # noqa: F401
