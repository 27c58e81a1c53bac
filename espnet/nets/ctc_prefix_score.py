"""espnet.nets.ctc_prefix_score (reference import path) -> auto_avsr_amd.ctc_prefix_score."""
from auto_avsr_amd.ctc_prefix_score import CTCPrefixScore, CTCPrefixScoreTH  # noqa: F401
