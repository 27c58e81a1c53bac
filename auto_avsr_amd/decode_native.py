"""Host side of csrc/decode.hip: the hybrid CTC / attention beam search with ONE library call per decoding step.

`NativeBeam.search` is what `decoding.BatchBeamSearch.forward` runs when the scorers are the ones the reference wires
(lightning.py:126-158: this build's TransformerDecoder, CTCPrefixScorer, optionally LengthBonus, pre-beam on the decoder
scores): same search (batch_beam_search.py:208-349 over beam_search.py:330-457), same hypotheses, with the per-step work --
decoder pass on the new position over cached K / V, pre-beam, CTC prefix scores, top-k, beam re-ordering -- issued from C++
(`avsr_beam_step`) instead of ~120 python-issued launches.  The python loop here only looks at the 1 KB the step copies
back (tokens, parents, scores), collects ended hypotheses and applies the end-detection rule.  Any other scorer
configuration keeps the python step of decoding.py."""
import contextlib
import ctypes
import math
import threading

import numpy as np
import torch

from . import _lib, ops

MAX_BEAM = 128


def _f32(t):
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class NativeBeam:
    """One session object per BatchBeamSearch: weights are re-bound when any decoder parameter changes."""

    def __init__(self, bs):
        self.bs = bs
        self.handle = 0
        self.key = None
        self.keep_alive = []

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().call("avsr_beam_destroy", self.handle)
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------ eligibility
    @staticmethod
    def supported(bs):
        from .decoding import CTCPrefixScorer, LengthBonus
        from .nets import TransformerDecoder

        full, part = bs.full_scorers, bs.part_scorers
        if set(part) != {"ctc"} or not isinstance(part["ctc"], CTCPrefixScorer):
            return False
        if "decoder" not in full or not isinstance(full["decoder"], TransformerDecoder):
            return False
        if set(full) - {"decoder", "length_bonus"}:
            return False
        if "length_bonus" in full and not isinstance(full["length_bonus"], LengthBonus):
            return False
        if not bs.do_pre_beam or bs.pre_beam_score_key not in ("decoder", "full") or bs.weights["decoder"] <= 0:
            return False
        dec = full["decoder"]
        lay = dec.decoders[0]
        D, H = lay.size, lay.self_attn.h
        FF = lay.feed_forward.w_1.out_features
        if dec.output_layer is None or not dec.normalize_before or D != 64 * H or FF % 64:
            return False

        def block_ok(nw):  # the block sizes csrc/decode.hip instantiates (waves of 64 columns of K)
            return 1 <= nw <= 12 and (nw <= 4 or nw % 2 == 0)

        if not block_ok(D // 64) or not any(FF % (64 * z) == 0 and block_ok(FF // (64 * z)) for z in range(1, 9)):
            return False
        beam, S = bs.beam_size, bs.pre_beam_size
        return 2 <= beam <= MAX_BEAM and beam <= S - 1 and beam * (S + 1) <= 8192

    # ------------------------------------------------------------------------------------------------ weights
    def _bind(self, dev, min_pos):
        bs = self.bs
        dec = bs.full_scorers["decoder"]
        params = list(dec.parameters())
        pos = dec.embed[1]
        pe = _f32(pos.table(max(pos.pe.size(1), min_pos), dev))
        # (FusedAdamW updates parameters through raw pointers without bumping `_version`: the optimizer-step generation of the
        # weight caches is part of the key, so the stacked copies below are rebuilt after native training steps too)
        from . import functional as AF

        key = (str(dev), pe.data_ptr(), pe.shape[0], AF._wgen["gen"]) + tuple((p.data_ptr(), p._version) for p in params)
        if key == self.key:
            return
        L = _lib.lib()
        if self.handle:
            L.call("avsr_beam_destroy", self.handle)
            self.handle = 0
        lay0 = dec.decoders[0]
        D, H, FF = lay0.size, lay0.self_attn.h, lay0.feed_forward.w_1.out_features
        emb = dec.embed[0]
        V = dec.output_layer.out_features
        shared = getattr(bs, "_native_weights", None)  # the stacked / f32 weight tensors: one set for all sessions of this search
        if shared is not None and shared[0] == key:
            keep = shared[1]
        else:
            keep = [_f32(emb.weight), pe]
            for d in dec.decoders:
                sa, ca, ff = d.self_attn, d.src_attn, d.feed_forward
                keep += [_f32(d.norm1.weight), _f32(d.norm1.bias),
                         torch.cat([_f32(sa.linear_q.weight), _f32(sa.linear_k.weight), _f32(sa.linear_v.weight)], 0).contiguous(),
                         torch.cat([_f32(sa.linear_q.bias), _f32(sa.linear_k.bias), _f32(sa.linear_v.bias)], 0).contiguous(),
                         _f32(sa.linear_out.weight), _f32(sa.linear_out.bias),
                         _f32(d.norm2.weight), _f32(d.norm2.bias), _f32(ca.linear_q.weight), _f32(ca.linear_q.bias),
                         torch.cat([_f32(ca.linear_k.weight), _f32(ca.linear_v.weight)], 0).contiguous(),
                         torch.cat([_f32(ca.linear_k.bias), _f32(ca.linear_v.bias)], 0).contiguous(),
                         _f32(ca.linear_out.weight), _f32(ca.linear_out.bias),
                         _f32(d.norm3.weight), _f32(d.norm3.bias), _f32(ff.w_1.weight), _f32(ff.w_1.bias), _f32(ff.w_2.weight),
                         _f32(ff.w_2.bias)]
            keep += [_f32(dec.after_norm.weight), _f32(dec.after_norm.bias), _f32(dec.output_layer.weight), _f32(dec.output_layer.bias)]
            bs._native_weights = (key, keep)
        assert all(t.device == keep[0].device for t in keep)
        has_len = int("length_bonus" in bs.full_scorers)
        cfg = (ctypes.c_int32 * 12)(D, H, FF, V, len(dec.decoders), bs.beam_size, bs.pre_beam_size, bs.sos, bs.eos,
                                    bs.part_scorers["ctc"].blank, has_len, pe.shape[0])
        fcfg = (ctypes.c_float * 5)(bs.weights["decoder"], bs.weights["ctc"], bs.weights.get("length_bonus", 0.0) if has_len else 0.0,
                                    pos.xscale, dec.decoders[0].norm1.eps)
        ptrs = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
        h = L.call("avsr_beam_create", ctypes.cast(cfg, ctypes.c_void_p), ctypes.cast(fcfg, ctypes.c_void_p),
                   ctypes.cast(ptrs, ctypes.c_void_p), len(keep))
        if not h:
            raise _lib.AvsrLibraryError("avsr_beam_create: " + L.cdll.avsr_last_error().decode())
        self.handle, self.key, self.keep_alive = h, key, keep
        self.V, self.pe_rows = V, pe.shape[0]
        pin = dev.type == "cuda"
        self.host = torch.empty(MAX_BEAM * 8, dtype=torch.float32, pin_memory=pin)
        self.host_np = self.host.numpy().reshape(MAX_BEAM, 8)
        self.yseq_host = None

    # ------------------------------------------------------------------------------------------------ the search
    @torch.no_grad()
    def search(self, x, maxlenratio=0.0, minlenratio=0.0, ctc_state=None):
        """ctc_state: (log-posteriors [T][ld], empty-prefix state [T][2]) from `prepare_ctc` when the caller computed them (search_many)."""
        from .decoding import Hypothesis

        bs = self.bs
        if maxlenratio == 0:
            maxlen = x.shape[0]
        elif maxlenratio < 0:
            maxlen = -1 * int(maxlenratio)
        else:
            maxlen = max(1, int(maxlenratio * x.size(0)))
        dev = x.device
        dec, ctc = bs.full_scorers["decoder"], bs.part_scorers["ctc"]
        self._bind(dev, maxlen + 2)
        L = _lib.lib()
        T = x.shape[0]
        logp, r_init = ctc_state if ctc_state is not None else prepare_ctc(ctc, x)
        memory = _f32(x)
        nws = L.call("avsr_beam_workspace_bytes", self.handle, T, maxlen)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        stream = ops._stream(memory)
        L.call("avsr_beam_begin", self.handle, memory.data_ptr(), T, logp.data_ptr(), logp.stride(0), r_init.data_ptr(), ws.data_ptr(),
               nws, maxlen, stream)
        n_out = ctypes.c_int(0)
        host_ptr, n_ptr = self.host.data_ptr(), ctypes.cast(ctypes.pointer(n_out), ctypes.c_void_p)
        names = ["decoder"] + (["length_bonus"] if "length_bonus" in bs.full_scorers else []) + ["ctc"]
        col = {"decoder": 3, "ctc": 4, "length_bonus": 5}
        eos = bs.eos
        ended, best, best_len = [], -math.inf, {}

        def fetch():
            yseq_host = self.yseq_host
            if yseq_host is None or yseq_host.shape[0] < bs.beam_size * (maxlen + 2):
                yseq_host = self.yseq_host = torch.empty(bs.beam_size * (maxlen + 2), dtype=torch.int64, pin_memory=dev.type == "cuda")
            ldy, Lc = ctypes.c_int(0), ctypes.c_int(0)
            L.call("avsr_beam_fetch_yseq", self.handle, yseq_host.data_ptr(), ctypes.cast(ctypes.pointer(ldy), ctypes.c_void_p),
                   ctypes.cast(ctypes.pointer(Lc), ctypes.c_void_p), stream)
            return yseq_host.numpy(), ldy.value, Lc.value

        def end(row, ys, forced):
            nonlocal best
            ys = list(ys) + ([eos] if forced else [])
            sc = float(row[2])
            ended.append(Hypothesis(yseq=torch.tensor(ys, dtype=torch.int64), score=sc, scores={k: float(row[col[k]]) for k in names},
                                    states={}))
            best = max(best, sc)
            best_len[len(ys)] = max(best_len.get(len(ys), -math.inf), sc)

        for i in range(maxlen):
            L.call("avsr_beam_step", self.handle, host_ptr, n_ptr, stream)
            K = n_out.value
            out = self.host_np[:K]
            tok = out[:, 0].astype(np.int64)
            if i == maxlen - 1:  # force an end so that at least one hypothesis finishes (beam_search.py:430-436)
                ys, ldy, Lc = fetch()
                for b in range(K):
                    end(out[b], ys[b * ldy: b * ldy + Lc], True)
                n_alive = 0
            else:
                is_eos = tok == eos
                n_alive = K
                if is_eos.any():
                    ys, ldy, Lc = fetch()
                    for b in np.nonzero(is_eos)[0].tolist():
                        end(out[b], ys[b * ldy: b * ldy + Lc], False)
                    keep = np.nonzero(~is_eos)[0].astype(np.int32)
                    n_alive = int(keep.shape[0])
                    L.call("avsr_beam_keep", self.handle, keep.ctypes.data, n_alive, stream)
            if maxlenratio == 0.0 and ended and self._end_detect(best, best_len, i):
                break
            if n_alive == 0:
                break
        nbest = sorted(ended, key=lambda h: float(h.score), reverse=True)
        if not nbest:
            return [] if minlenratio < 0.1 else self.search(x, maxlenratio, max(0.0, minlenratio - 0.1), ctc_state=(logp, r_init))
        return nbest

    @staticmethod
    def _end_detect(best, best_len, i, M=3, D_end=math.log(1 * math.exp(-10))):
        """decoding.end_detect (e2e_asr_common.py:17-47) on the running maxima instead of the list of dicts."""
        count = 0
        for m in range(M):
            s = best_len.get(i - m)
            if s is not None and s - best < D_end:
                count += 1
        return count == M


def prepare_ctc(ctc_scorer, x):
    """CTC log-posteriors [T][ld] of one utterance and the state of the empty prefix [T][2] (scorers/ctc.py:87-99)."""
    r0, _ = ctc_scorer.batch_init_state(x)
    return ctc_scorer.logp, r0.reshape(x.shape[0], 2).contiguous()


def search_many(bs, xs, workers=4, maxlenratio=0.0, minlenratio=0.0):
    """Beam searches of several utterances AT ONCE: `workers` host threads, each with its own session (beam state, workspace,
    pinned result buffer) and its own stream.  A decoding step is 52 dependent launches of 24 - 316 blocks that are bound by
    launch / memory latency, not by the machine: steps of different utterances overlap on the GPU, and the library call
    releases the GIL while it issues the launches and waits for the step's result.  Results are those of `bs(x)` per utterance
    (each search is the same sequence of launches on its own state).  The CTC posteriors are computed here, on the caller's
    thread: the kernels' python glue (functional.py mode state, weight caches) is not re-entrant."""
    if not xs:
        return []
    dev = xs[0].device
    cuda = dev.type == "cuda"
    ctc = bs.part_scorers["ctc"]
    pre = [prepare_ctc(ctc, x) for x in xs]
    maxpos = max((x.shape[0] if maxlenratio == 0 else (-int(maxlenratio) if maxlenratio < 0 else max(1, int(maxlenratio * x.shape[0]))))
                 for x in xs) + 2
    workers = max(1, min(workers, len(xs)))
    pool = getattr(bs, "_native_pool", None)
    if pool is None or len(pool) < workers:
        pool = bs._native_pool = (pool or []) + [NativeBeam(bs) for _ in range(workers - len(pool or []))]
    for sess in pool[:workers]:
        sess._bind(dev, maxpos)  # (on this thread: binding extends the decoder's position table)
    streams = [torch.cuda.Stream(device=dev) for _ in range(workers)] if cuda else [None] * workers
    if cuda:
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))  # encoder outputs / posteriors were produced on the caller's stream
    results, errors, it, lock = [None] * len(xs), [], iter(range(len(xs))), threading.Lock()

    def work(w):
        ctx = torch.cuda.stream(streams[w]) if cuda else contextlib.nullcontext()
        try:
            with ctx, torch.no_grad():
                while True:
                    with lock:
                        i = next(it, None)
                    if i is None or errors:
                        break
                    results[i] = pool[w].search(xs[i], maxlenratio, minlenratio, ctc_state=pre[i])
        except BaseException as e:  # noqa: BLE001 -- re-raised on the caller's thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(w,), daemon=True) for w in range(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if cuda:
        for st in streams:
            torch.cuda.current_stream(dev).wait_stream(st)
    if errors:
        raise errors[0]
    return results
