"""espnet.nets.scorers.ctc (reference import path) -> auto_avsr_amd.decoding."""
from auto_avsr_amd.decoding import CTCPrefixScorer  # noqa: F401
