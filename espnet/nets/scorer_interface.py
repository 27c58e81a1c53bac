"""espnet.nets.scorer_interface (reference import path) -> auto_avsr_amd.scorer_interface."""
from auto_avsr_amd.scorer_interface import (  # noqa: F401
    BatchPartialScorerInterface, BatchScorerInterface, PartialScorerInterface, ScorerInterface)
