"""CTC prefix scoring for hybrid CTC / attention beam search (espnet/nets/ctc_prefix_score.py).

`CTCPrefixScoreTH` keeps the reference's constructor, call and state contract (ctc_prefix_score.py:10-219) but runs
the frame recursion -- the python `for t in range(start, end)` loop of :155-160, ~T small launches per decoding step
-- as ONE launch of `avsr_ctc_prefix_score` (csrc/ctc_prefix.hip: one thread per (hypothesis, candidate) pair walks
the T frames with its forward variables in registers).  `CTCPrefixScore` is the single-hypothesis host form
(:264-357, numpy) that the non-batched `CTCPrefixScorer.score_partial` uses.

State layout (unchanged from the reference): r [T, 2, n_bh(, S)] = log forward probabilities of a prefix ending in a
non-blank (index 0) / blank (index 1) at frame t; log_psi [n_bh, O] = log prefix probabilities.
"""
import numpy as np
import torch

from . import ops

LOGZERO = -10000000000.0


class CTCPrefixScoreTH(object):
    """Batched prefix scorer: B utterances x W hypotheses each (Watanabe et al. Algorithm 2, vectorised as in Seki et
    al. 2019).  x (B, T, O) log posteriors, xlens (B,), blank / eos ids.  margin > 0 (attention-windowed scoring,
    ctc_prefix_score.py:139-150) is not wired by any caller in the reference and is refused."""

    def __init__(self, x, xlens, blank, eos, margin=0):
        if margin != 0:
            raise NotImplementedError("CTCPrefixScoreTH: windowed scoring (margin > 0) is not supported")
        self.logzero = LOGZERO
        self.blank, self.eos, self.margin = blank, eos, margin
        self.batch, self.input_length, self.odim = x.size(0), x.size(1), x.size(2)
        self.dtype = x.dtype
        self.device = x.device
        xlens = [int(v) for v in (xlens.tolist() if torch.is_tensor(xlens) else xlens)]
        # frames past an utterance's end emit blank with probability 1 (ctc_prefix_score.py:45-50)
        logp = x.detach().to(torch.float32).clone()
        for i, n in enumerate(xlens):
            if n < self.input_length:
                logp[i, n:, :] = self.logzero
                logp[i, n:, blank] = 0
        self.logp = logp.contiguous()  # (B, T, O): one [T][O] matrix per utterance for the kernel
        self.end_frames = torch.as_tensor(xlens) - 1
        self.idx_b = torch.arange(self.batch, device=self.device)
        self.scoring_num = 0

    @property
    def x(self):
        """The reference's (2, T, B, O) view {posteriors, blank posterior broadcast over O} (ctc_prefix_score.py:52-54)."""
        xn = self.logp.transpose(0, 1)
        return torch.stack([xn, xn[:, :, self.blank].unsqueeze(2).expand(-1, -1, self.odim)])

    def __call__(self, y, state, scoring_ids=None, att_w=None):
        """y: n_bh prefixes (each starting with sos); state: None or (r_prev [T, 2, n_bh], s_prev, f_min, f_max);
        scoring_ids: (n_bh, S) candidate tokens or None (score the whole vocabulary).
        Returns (log_psi - s_prev) [n_bh, O] and the new state (r, log_psi, f_min, f_max, scoring_idmap)."""
        output_length = len(y[0]) - 1
        last_ids = torch.as_tensor([int(yi[-1]) for yi in y], dtype=torch.int64, device=self.device)
        n_bh = len(y)
        n_hyps = n_bh // self.batch
        T, O, dev = self.input_length, self.odim, self.device
        self.scoring_num = scoring_ids.size(-1) if scoring_ids is not None else 0
        if state is None:
            r_prev = torch.full((T, 2, self.batch, n_hyps), self.logzero, dtype=torch.float32, device=dev)
            r_prev[:, 1] = torch.cumsum(self.logp[:, :, self.blank].transpose(0, 1), 0).unsqueeze(2)
            r_prev = r_prev.view(T, 2, n_bh)
            s_prev, f_min_prev, f_max_prev = 0.0, 0, 1
        else:
            r_prev, s_prev, f_min_prev, f_max_prev = state
            r_prev = r_prev.to(torch.float32)
        if self.scoring_num > 0:
            cand = scoring_ids.to(torch.int64).contiguous()
            snum = self.scoring_num
            scoring_idmap = torch.full((n_bh, O), -1, dtype=torch.long, device=dev)
            scoring_idmap.scatter_(1, cand, torch.arange(snum, device=dev).unsqueeze(0).expand(n_bh, snum))
        else:
            cand = torch.arange(O, dtype=torch.int64, device=dev).unsqueeze(0).expand(n_bh, O).contiguous()
            snum = O
            scoring_idmap = None
        r = torch.empty(T, 2, n_bh, snum, dtype=torch.float32, device=dev)
        psi = torch.empty(n_bh, snum, dtype=torch.float32, device=dev)
        psi_eos = torch.empty(n_bh, dtype=torch.float32, device=dev)
        for b in range(self.batch):
            lo, hi = b * n_hyps, (b + 1) * n_hyps
            if self.batch == 1:
                rp, rn, ps, pe = r_prev.contiguous(), r, psi, psi_eos
            else:
                rp = r_prev[:, :, lo:hi].contiguous()
                rn = torch.empty(T, 2, n_hyps, snum, dtype=torch.float32, device=dev)
                ps, pe = psi[lo:hi], psi_eos[lo:hi]
            lg = self.logp[b]
            ops.call("avsr_ctc_prefix_score", ops._ptr(lg), T, O, lg.stride(0), ops._ptr(rp),
                     ops._ptr(last_ids[lo:hi].contiguous()), ops._ptr(cand[lo:hi].contiguous()), n_hyps, snum,
                     output_length, self.blank, ops._ptr(rn), ops._ptr(ps), ops._ptr(pe), ops._stream(lg))
            if self.batch != 1:
                r[:, :, lo:hi] = rn
        if scoring_idmap is not None:
            log_psi = torch.full((n_bh, O), self.logzero, dtype=torch.float32, device=dev)
            log_psi.scatter_(1, cand, psi)
        else:
            log_psi = psi
        # P(prefix is the complete label sequence): forward probability at each utterance's last real frame
        r_sum = torch.logsumexp(r_prev, 1)  # (T, n_bh)
        ends = self.end_frames.to(dev).repeat_interleave(n_hyps)
        log_psi[:, self.eos] = r_sum[ends, torch.arange(n_bh, device=dev)]
        log_psi[:, self.blank] = self.logzero
        out = log_psi.to(self.dtype)
        return (out - s_prev), (r.to(self.dtype), out, 0, 0, scoring_idmap)

    def index_select_state(self, state, best_ids):
        """State of the pruned beam: best_ids (B, W) index the flattened (hypothesis, token) space of each utterance
        (ctc_prefix_score.py:189-219)."""
        r, s, f_min, f_max, scoring_idmap = state
        n_bh = len(s)
        n_hyps = n_bh // self.batch
        vidx = (best_ids + (self.idx_b * (n_hyps * self.odim)).view(-1, 1)).view(-1)
        s_new = torch.index_select(s.reshape(-1), 0, vidx).view(-1, 1).repeat(1, self.odim).view(n_bh, self.odim)
        if scoring_idmap is not None:
            snum = self.scoring_num
            hyp_idx = (torch.div(best_ids, self.odim, rounding_mode="floor") + (self.idx_b * n_hyps).view(-1, 1)).view(-1)
            label_ids = torch.fmod(best_ids, self.odim).view(-1)
            score_idx = scoring_idmap[hyp_idx, label_ids]
            score_idx = torch.where(score_idx == -1, torch.zeros_like(score_idx), score_idx)
            vidx = score_idx + hyp_idx * snum
        else:
            snum = self.odim
        r_new = torch.index_select(r.reshape(-1, 2, n_bh * snum), 2, vidx).view(-1, 2, n_bh)
        return r_new, s_new, f_min, f_max

    def extend_prob(self, x):
        """Streaming decoding (ctc_prefix_score.py:221-241): longer posteriors for the same (single) utterance."""
        if self.input_length < x.shape[1]:
            new = x.detach().to(torch.float32).clone()
            new[:, : self.input_length] = self.logp
            self.logp = new.contiguous()
            self.input_length = x.size(1)
            self.end_frames = torch.as_tensor([x.size(1)]) - 1

    def extend_state(self, state):
        """Continue a hypothesis's blank path over the newly appended frames (ctc_prefix_score.py:243-262)."""
        if state is None:
            return state
        r_prev, s_prev, f_min_prev, f_max_prev = state
        r_new = torch.full((self.input_length, 2), self.logzero, dtype=self.dtype, device=self.device)
        start = max(r_prev.shape[0], 1)
        r_new[0:start] = r_prev
        if start < self.input_length:
            tail = torch.cumsum(self.logp[0, start:, self.blank], 0).to(self.dtype)
            r_new[start:, 1] = r_new[start - 1, 1] + tail
        return r_new, s_prev, f_min_prev, f_max_prev


class CTCPrefixScore(object):
    """Single-hypothesis host form (ctc_prefix_score.py:264-357): x [T, O] numpy log posteriors; `xp` is the array module
    (numpy).  Used by CTCPrefixScorer.init_state / score_partial, i.e. the non-batched BeamSearch."""

    def __init__(self, x, blank, eos, xp=np):
        self.xp, self.logzero = xp, LOGZERO
        self.blank, self.eos = blank, eos
        self.input_length = len(x)
        self.x = x

    def initial_state(self):
        """r [T, 2]: only blanks so far."""
        r = self.xp.full((self.input_length, 2), self.logzero, dtype=np.float32)
        r[:, 1] = self.xp.cumsum(self.x[:, self.blank].astype(np.float64)).astype(np.float32)
        return r

    def __call__(self, y, cs, r_prev):
        """(log prefix probabilities of y + c for c in cs, new states [len(cs), T, 2])."""
        xp = self.xp
        cs = np.asarray(cs)
        out_len = len(y) - 1
        T = self.input_length
        r = xp.full((T, 2, len(cs)), self.logzero, dtype=np.float32)
        xs = self.x[:, cs]
        if out_len == 0:
            r[0, 0] = xs[0]
        r_sum = xp.logaddexp(r_prev[:, 0], r_prev[:, 1])
        last = int(y[-1])
        log_phi = xp.repeat(r_sum[:, None], len(cs), axis=1)
        if out_len > 0:
            log_phi[:, cs == last] = r_prev[:, 1][:, None]
        start = max(out_len, 1)
        log_psi = r[start - 1, 0].copy()
        for t in range(start, T):
            r[t, 0] = xp.logaddexp(r[t - 1, 0], log_phi[t - 1]) + xs[t]
            r[t, 1] = xp.logaddexp(r[t - 1, 0], r[t - 1, 1]) + self.x[t, self.blank]
            log_psi = xp.logaddexp(log_psi, log_phi[t - 1] + xs[t])
        log_psi[cs == self.eos] = r_sum[-1]
        log_psi[cs == self.blank] = self.logzero
        return log_psi, xp.rollaxis(r, 2)
