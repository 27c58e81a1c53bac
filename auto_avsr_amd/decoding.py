"""Hybrid CTC / attention beam search for evaluation.

Behavioural contract = the reference's decoding stack as `lightning.get_beam_search_decoder` wires it
(lightning.py:126-158): `BatchBeamSearch` (espnet/nets/batch_beam_search.py:26-349 on top of beam_search.py:30-457)
with the scorers {"decoder": TransformerDecoder (full, weight 1 - ctc_weight), "ctc": CTCPrefixScorer (partial, weight
ctc_weight), "length_bonus": LengthBonus (weight `penalty`)}, beam 40, pre-beam int(1.5 * beam) on the decoder scores,
end detection of e2e_asr_common.py:17-47, hypotheses as `Hypothesis(yseq, score, scores, states)` with `.asdict()`.

Design (not a transliteration): the running beam is kept as batched tensors for the whole search -- token matrix,
scores, one decoder cache tensor per layer, one CTC state tensor -- so a decoding step is: one incremental decoder pass
over all hypotheses (existing kernels, KV cache = previous layer outputs), one top-k for the pre-beam, ONE launch of the
CTC prefix-score kernel (csrc/ctc_prefix.hip: the reference loops over the T frames in python, ~400 small launches per
step), one top-k over beam x vocabulary, and index_select gathers.  Nothing is unbatched into per-hypothesis python
objects until hypotheses end."""
import math
import os
from typing import Any, Dict, List, NamedTuple, Optional

import torch

from . import ops
from .ctc_prefix_score import CTCPrefixScore, CTCPrefixScoreTH
from .scorer_interface import BatchPartialScorerInterface, BatchScorerInterface

LOGZERO = -10000000000.0
# A/B switch: 0 keeps the python-issued decoding step below for every scorer configuration (default: the reference's scorer set
# runs the one-call-per-step search of decode_native.py / csrc/decode.hip)
NATIVE_BEAM = os.environ.get("AVSR_NATIVE_BEAM", "1") != "0"


class Hypothesis(NamedTuple):
    """beam_search.py:13-27."""

    yseq: torch.Tensor
    score: Any = 0.0
    scores: Dict[str, Any] = dict()
    states: Dict[str, Any] = dict()

    def asdict(self) -> dict:
        return self._replace(yseq=self.yseq.tolist(), score=float(self.score),
                             scores={k: float(v) for k, v in self.scores.items()})._asdict()


def end_detect(ended_hyps, i, M=3, D_end=math.log(1 * math.exp(-10))):
    """e2e_asr_common.py:17-47 (Eq. 50 of Watanabe et al.): stop when, for each of the last M lengths, the best ended
    hypothesis of that length scores more than |D_end| below the best ended hypothesis overall."""
    if not ended_hyps:
        return False
    best = max(h["score"] for h in ended_hyps)
    count = 0
    for m in range(M):
        same = [h["score"] for h in ended_hyps if len(h["yseq"]) == i - m]
        if same and max(same) - best < D_end:
            count += 1
    return count == M


class LengthBonus(BatchScorerInterface):
    """scorers/length_bonus.py:10-59: a constant 1 per emitted token."""

    def __init__(self, n_vocab: int):
        self.n = n_vocab

    def batch_init_state(self, x):
        return None

    def batch_score(self, ys, states, xs):
        return torch.ones(1, dtype=xs.dtype, device=xs.device).expand(ys.shape[0], self.n), None

    def score(self, y, state, x):
        return torch.ones(1, dtype=x.dtype, device=x.device).expand(self.n), None


class CTCPrefixScorer(BatchPartialScorerInterface):
    """scorers/ctc.py:10-157 over ctc_prefix_score.py, for one utterance.  Two state contracts are served:

    * the tensor contract this build's BatchBeamSearch drives (batch_init_state returns the state of the empty
      prefix; batch_score_partial takes / returns whole-beam tensors; select_states gathers the new beam) -- state of
      a beam of n hypotheses: (r [T, 2, n], s [n]) = forward variables of each prefix (ending in non-blank / blank)
      and its log prefix probability;
    * the reference's own contract, for callers written against it (the reference's BeamSearch / BatchBeamSearch):
      init_state / score_partial / select_state on the host form `CTCPrefixScore` (scorers/ctc.py:26-85) and
      batch_score_partial on a LIST of per-hypothesis states through `self.impl` = `CTCPrefixScoreTH`
      (scorers/ctc.py:87-126), selected by the type of `state`."""

    def __init__(self, ctc: torch.nn.Module, eos: int):
        self.ctc = ctc
        self.eos = eos
        self.blank = 0
        self.logp = None
        self.impl = None

    # -- reference contract, single hypothesis (scorers/ctc.py:26-85)
    def init_state(self, x: torch.Tensor):
        logp = self.ctc.log_softmax(x.unsqueeze(0)).detach().squeeze(0).float().cpu().numpy()
        self.impl = CTCPrefixScore(logp, 0, self.eos)
        return 0, self.impl.initial_state()

    def select_state(self, state, i, new_id=None):
        if type(state) == tuple:
            if len(state) == 2:  # CTCPrefixScore: (scores, states) indexed by candidate
                sc, st = state
                return sc[i], st[i]
            r, log_psi, f_min, f_max, scoring_idmap = state  # CTCPrefixScoreTH
            s = log_psi[i, new_id].expand(log_psi.size(1))
            col = scoring_idmap[i, new_id] if scoring_idmap is not None else new_id
            return r[:, :, i, col], s, f_min, f_max
        return None if state is None else state[i]

    def score_partial(self, y, ids, state, x):
        prev_score, st = state
        presub, new_st = self.impl(y.cpu(), ids.cpu(), st)
        return torch.as_tensor(presub - prev_score, device=x.device, dtype=x.dtype), (presub, new_st)

    def extend_prob(self, x: torch.Tensor):
        self.impl.extend_prob(self.ctc.log_softmax(x.unsqueeze(0)))

    def extend_state(self, state):
        return [self.impl.extend_state(s) for s in state]

    def batch_init_state(self, x: torch.Tensor):
        logp = self.ctc.log_softmax(x.unsqueeze(0)).detach().squeeze(0)  # (T, V[, pad])
        self.logp = logp.float().contiguous()
        self.T = logp.shape[0]
        self.odim = self.ctc.ctc_lo.out_features
        # the reference's batch implementation object (scorers/ctc.py:96-98), for callers that use its list-of-states
        # contract; shares nothing mutable with the tensor contract below
        self.impl = CTCPrefixScoreTH(self.logp[:, : self.odim].unsqueeze(0), torch.tensor([self.T]), 0, self.eos)
        r0 = torch.full((self.T, 2, 1), LOGZERO, dtype=torch.float32, device=x.device)
        r0[:, 1, 0] = torch.cumsum(self.logp[:, self.blank], 0)  # only blanks so far
        return r0, torch.zeros(1, dtype=torch.float32, device=x.device)

    def batch_score_partial(self, yseq: torch.Tensor, ids: torch.Tensor, state, x=None):
        """yseq [n, L] (sos first), ids [n, S] candidate tokens.  Returns (scores [n, V]: log psi(prefix + v) - log
        psi(prefix), LOGZERO outside the candidates / for blank, the complete-sequence probability for eos) and the
        state of every (hypothesis, candidate): (r_new [T, 2, n, S], log_psi [n, V], ids)."""
        if state is None or isinstance(state, list):  # the reference's contract (scorers/ctc.py:101-126)
            batch_state = None if (state is None or state[0] is None) else (
                torch.stack([st[0] for st in state], dim=2), torch.stack([st[1] for st in state]), state[0][2], state[0][3])
            return self.impl(yseq, batch_state, ids)
        r_prev, s_prev = state
        n, S = ids.shape
        dev = ids.device
        r_new = torch.empty(self.T, 2, n, S, dtype=torch.float32, device=dev)
        psi = torch.empty(n, S, dtype=torch.float32, device=dev)
        psi_eos = torch.empty(n, dtype=torch.float32, device=dev)
        last = yseq[:, -1].contiguous()
        ids = ids.contiguous()
        ops.call("avsr_ctc_prefix_score", ops._ptr(self.logp), self.T, self.odim, self.logp.stride(0),
                 ops._ptr(r_prev.contiguous()), ops._ptr(last), ops._ptr(ids), n, S, yseq.shape[1] - 1, self.blank,
                 ops._ptr(r_new), ops._ptr(psi), ops._ptr(psi_eos), ops._stream(self.logp))
        log_psi = torch.full((n, self.odim), LOGZERO, dtype=torch.float32, device=dev)
        log_psi.scatter_(1, ids, psi)
        log_psi[:, self.eos] = psi_eos
        log_psi[:, self.blank] = LOGZERO
        return log_psi - s_prev.unsqueeze(1), (r_new, log_psi, ids)

    @staticmethod
    def select_states(state, prev: torch.Tensor, tok: torch.Tensor):
        """State of the new beam: hypothesis prev[j] extended by tok[j]."""
        r_new, log_psi, ids = state
        hit = ids[prev] == tok.unsqueeze(1)  # position of the chosen token among its parent's candidates
        pos = torch.where(hit.any(1), hit.float().argmax(1), torch.full_like(tok, ids.shape[1] - 1))  # (eos may be absent)
        return r_new[:, :, prev, pos].contiguous(), log_psi[prev, tok]


class BatchBeamSearch(torch.nn.Module):
    """Constructor and call signature of batch_beam_search.BatchBeamSearch / beam_search.BeamSearch."""

    def __init__(self, scorers: Dict[str, Any], weights: Dict[str, float], beam_size: int, vocab_size: int, sos: int,
                 eos: int, token_list: Optional[List[str]] = None, pre_beam_ratio: float = 1.5,
                 pre_beam_score_key: Optional[str] = None):
        super().__init__()
        self.weights = weights
        self.scorers, self.full_scorers, self.part_scorers = {}, {}, {}
        for k, v in scorers.items():
            if weights.get(k, 0) == 0 or v is None:  # beam_search.py:75-78
                continue
            self.scorers[k] = v
            (self.part_scorers if hasattr(v, "batch_score_partial") else self.full_scorers)[k] = v
        self.sos, self.eos, self.token_list = sos, eos, token_list
        self.beam_size, self.n_vocab = beam_size, vocab_size
        self.pre_beam_size = int(pre_beam_ratio * beam_size)
        if pre_beam_score_key is not None and pre_beam_score_key != "full" and pre_beam_score_key not in self.full_scorers:
            raise KeyError(f"{pre_beam_score_key} is not found in {self.full_scorers}")
        self.pre_beam_score_key = pre_beam_score_key
        self.do_pre_beam = (pre_beam_score_key is not None and self.pre_beam_size < self.n_vocab
                            and len(self.part_scorers) > 0)
        self._native = None  # decode_native.NativeBeam, False = this scorer set stays on the python step

    # ---------------------------------------------------------------------------------------------- one step
    def _step(self, beam, x):
        """beam: dict(yseq [n, L], score [n], scores {k: [n]}, states {k: batched state}) -> the `beam_size` best
        one-token extensions, same structure."""
        yseq, n = beam["yseq"], beam["yseq"].shape[0]
        V = self.n_vocab
        weighted = torch.zeros(n, V, dtype=torch.float32, device=x.device)
        full_scores, full_states = {}, {}
        xs = x.unsqueeze(0).expand(n, *x.shape)
        for k, d in self.full_scorers.items():
            if k == "decoder" or hasattr(d, "forward_one_step"):
                sc, st = self._decoder_step(d, yseq, beam["states"][k], xs)
            else:
                sc, st = d.batch_score(yseq, beam["states"][k], xs)
            full_scores[k], full_states[k] = sc[:, :V].float(), st
            weighted += self.weights[k] * full_scores[k]
        part_scores, part_states = {}, {}
        if self.part_scorers:
            if self.do_pre_beam:
                pre = weighted if self.pre_beam_score_key == "full" else full_scores[self.pre_beam_score_key]
                part_ids = torch.topk(pre, self.pre_beam_size, dim=-1)[1]
            else:
                part_ids = torch.arange(V, device=x.device).unsqueeze(0).expand(n, V)
            for k, d in self.part_scorers.items():
                part_scores[k], part_states[k] = d.batch_score_partial(yseq, part_ids, beam["states"][k], x)
                weighted += self.weights[k] * part_scores[k]
        weighted += beam["score"].unsqueeze(1)
        top = weighted.view(-1).topk(min(self.beam_size, weighted.numel()))[1]
        prev, tok = torch.div(top, V, rounding_mode="trunc"), top % V
        new = {"yseq": torch.cat([yseq[prev], tok.unsqueeze(1)], 1), "score": weighted[prev, tok], "scores": {},
               "states": {}}
        for k in self.full_scorers:
            new["scores"][k] = beam["scores"][k][prev] + full_scores[k][prev, tok]
            st = full_states[k]
            new["states"][k] = None if st is None else [c[prev] for c in st]
        for k, d in self.part_scorers.items():
            new["scores"][k] = beam["scores"][k][prev] + part_scores[k][prev, tok]
            new["states"][k] = d.select_states(part_states[k], prev, tok)
        return new

    @staticmethod
    def _decoder_step(dec, yseq, cache, xs):
        """TransformerDecoder.batch_score semantics (transformer_decoder.py:301-334) on batched cache tensors:
        log-probabilities of the next token for every hypothesis + the per-layer outputs of all positions so far."""
        from .nets import subsequent_mask

        mask = subsequent_mask(yseq.size(-1), device=xs.device).unsqueeze(0)
        logp, new_cache = dec.forward_one_step(yseq, mask, xs, cache=cache)
        return logp, new_cache

    # ---------------------------------------------------------------------------------------------- search loop
    @torch.no_grad()
    def forward(self, x: torch.Tensor, maxlenratio: float = 0.0, minlenratio: float = 0.0) -> List[Hypothesis]:
        """x: encoder output of one utterance (T, D).  Returns the ended hypotheses, best first (beam_search.py:330-406)."""
        if NATIVE_BEAM and self._native is not False:
            if self._native is None:
                from .decode_native import NativeBeam

                self._native = NativeBeam(self) if NativeBeam.supported(self) else False
            if self._native:
                return self._native.search(x, maxlenratio, minlenratio)
        if maxlenratio == 0:
            maxlen = x.shape[0]
        elif maxlenratio < 0:
            maxlen = -1 * int(maxlenratio)
        else:
            maxlen = max(1, int(maxlenratio * x.size(0)))
        dev = x.device
        beam = {"yseq": torch.full((1, 1), self.sos, dtype=torch.int64, device=dev),
                "score": torch.zeros(1, dtype=torch.float32, device=dev),
                "scores": {k: torch.zeros(1, dtype=torch.float32, device=dev) for k in self.scorers},
                "states": {k: d.batch_init_state(x) if hasattr(d, "batch_init_state") else None
                           for k, d in self.scorers.items()}}
        ended: List[Hypothesis] = []
        for i in range(maxlen):
            beam = self._step(beam, x)
            if i == maxlen - 1:  # force an end so that at least one hypothesis finishes (beam_search.py:430-436)
                beam["yseq"] = torch.cat([beam["yseq"], torch.full((beam["yseq"].shape[0], 1), self.eos, dtype=torch.int64,
                                                                     device=dev)], 1)
            is_eos = beam["yseq"][:, -1] == self.eos
            if bool(is_eos.any()):
                for b in torch.nonzero(is_eos).view(-1).tolist():
                    ended.append(Hypothesis(yseq=beam["yseq"][b].clone(), score=beam["score"][b],
                                            scores={k: v[b] for k, v in beam["scores"].items()}, states={}))
                keep = torch.nonzero(~is_eos).view(-1)
                beam = {"yseq": beam["yseq"][keep], "score": beam["score"][keep],
                        "scores": {k: v[keep] for k, v in beam["scores"].items()},
                        "states": {k: self._keep_state(k, st, keep) for k, st in beam["states"].items()}}
            if maxlenratio == 0.0 and end_detect([h.asdict() for h in ended], i):
                break
            if beam["yseq"].shape[0] == 0:
                break
        nbest = sorted(ended, key=lambda h: float(h.score), reverse=True)
        if not nbest:
            return [] if minlenratio < 0.1 else self.forward(x, maxlenratio, max(0.0, minlenratio - 0.1))
        return nbest

    def forward_many(self, xs, workers: int = 4, maxlenratio: float = 0.0, minlenratio: float = 0.0):
        """Not in the reference: the searches of several utterances (encoder outputs (T_i, D)) run concurrently -- one host thread,
        one stream and one session of csrc/decode.hip per worker (decode_native.search_many).  Returns [self(x) for x in xs];
        scorer sets the library does not cover are searched one after the other by the python step."""
        if NATIVE_BEAM and self._native is not False:
            if self._native is None:
                from .decode_native import NativeBeam

                self._native = NativeBeam(self) if NativeBeam.supported(self) else False
            if self._native:
                from .decode_native import search_many

                return search_many(self, list(xs), workers, maxlenratio, minlenratio)
        return [self(x, maxlenratio, minlenratio) for x in xs]

    def _keep_state(self, k, st, keep):
        if st is None:
            return None
        if k in self.part_scorers:
            r, s = st
            return r[:, :, keep].contiguous(), s[keep]
        return [c[keep] for c in st]
