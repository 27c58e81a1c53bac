"""espnet.nets.scorers.length_bonus (reference import path) -> auto_avsr_amd.decoding."""
from auto_avsr_amd.decoding import LengthBonus  # noqa: F401
