"""espnet.nets.beam_search (reference import path) -> auto_avsr_amd.decoding (the batched search is the only one)."""
from auto_avsr_amd.decoding import BatchBeamSearch as BeamSearch  # noqa: F401
from auto_avsr_amd.decoding import Hypothesis  # noqa: F401
