"""The four scorer base classes of the reference's beam search (espnet/nets/scorer_interface.py:9-186).

A *scorer* contributes one weighted term to the beam-search objective.  The hierarchy mirrors the reference so that
`isinstance(decoder, BatchScorerInterface)` / `isinstance(ctc_scorer, PartialScorerInterface)` -- the tests
beam_search.py:69-84 uses to sort scorers into full and partial ones -- keep working on this build's scorers:

    ScorerInterface                 score(y, state, x) over the whole vocabulary     (scorer_interface.py:9-79)
      BatchScorerInterface          batch_score(ys, states, xs)                      (:82-125)
      PartialScorerInterface        score_partial(y, next_tokens, state, x)          (:128-160)
        BatchPartialScorerInterface batch_score_partial(ys, next_tokens, states, xs) (:163-186, inherits both)

Only the state-handling defaults carry behaviour; they are restated here (not imported: the reference is not a
dependency of this package).
"""
import warnings
from typing import Any, List, Tuple

import torch


class ScorerInterface:
    """Scores every vocabulary entry as the next token of one hypothesis."""

    def init_state(self, x: torch.Tensor) -> Any:
        """State before the first token; `x` is the encoder output of the utterance.  Default: stateless."""
        return None

    def select_state(self, state: Any, i: int, new_id: int = None) -> Any:
        """State of hypothesis `i` of the beam (optionally specialised to the new token `new_id`)."""
        if state is None:
            return None
        return state[i]

    def score(self, y: torch.Tensor, state: Any, x: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        """(scores [n_vocab], next state) for the 1-D int64 prefix `y`.  Must be overridden."""
        raise NotImplementedError

    def final_score(self, state: Any) -> float:
        """Extra score when the hypothesis ends (eos).  Default: none."""
        return 0.0


class BatchScorerInterface(ScorerInterface):
    """Scorer that can process the whole beam at once."""

    def batch_init_state(self, x: torch.Tensor) -> Any:
        return self.init_state(x)

    def batch_score(self, ys: torch.Tensor, states: List[Any], xs: torch.Tensor) -> Tuple[torch.Tensor, List[Any]]:
        """(scores [n_batch, n_vocab], next states) for prefixes ys [n_batch, ylen].  The default walks the beam with
        `score` and says so (scorer_interface.py:112-125); batched scorers override it."""
        warnings.warn(f"{type(self).__name__} batch score is implemented through for loop not parallelized")
        rows, nxt = [], []
        for y, st, x in zip(ys, states, xs):
            sc, st2 = self.score(y, st, x)
            rows.append(sc)
            nxt.append(st2)
        return torch.cat(rows, 0).view(ys.shape[0], -1), nxt


class PartialScorerInterface(ScorerInterface):
    """Scorer that is too expensive to run on the whole vocabulary: it is given the pre-pruned candidate tokens."""

    def score_partial(self, y: torch.Tensor, next_tokens: torch.Tensor, state: Any,
                      x: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        """(scores [len(next_tokens)], next state).  Must be overridden."""
        raise NotImplementedError


class BatchPartialScorerInterface(BatchScorerInterface, PartialScorerInterface):
    """Partial scorer over the whole beam."""

    def batch_score_partial(self, ys: torch.Tensor, next_tokens: torch.Tensor, states: List[Any],
                            xs: torch.Tensor) -> Tuple[torch.Tensor, Any]:
        """(scores [n_batch, n_vocab], next states) for candidates next_tokens [n_batch, n_token].  Must be overridden."""
        raise NotImplementedError
