"""Decode throughput of the evaluation path (lightning.ModelModule.forward -- front-end, encoder, hybrid CTC /
attention BatchBeamSearch with beam 40) on the MI355X: utterances/s and ms per emitted token for T = 100 and T = 400 frames,
bf16 and precise numerical modes.  Weights: tests/golden/synth.py (the decode goldens' generator), so the search runs a
realistic number of steps instead of collapsing on an untrained model's first <eos>.
    python tools/bench_decode.py [--reps 3] > profiles/r3_decode_throughput.json
The reference's own CPU figure for the same loop: 2.12 s for one 4 s utterance (T = 100)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--beam", type=int, default=40)
    ap.add_argument("--many", type=int, default=16, help="utterances of the concurrent-decoding rows (BatchBeamSearch.forward_many)")
    args = ap.parse_args()
    import lightning
    from synth import synth_batch, synth_state_dict

    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.e2e import E2E

    dev = torch.device("cuda:0")
    m = E2E(5049, "video")
    m.load_state_dict(synth_state_dict(m.state_dict(), 3))
    m = m.to(dev).eval()
    bs = lightning.get_beam_search_decoder(m, [str(i) for i in range(5049)], beam_size=args.beam)
    rows = []
    from auto_avsr_amd import decoding

    for mode, native in (("mixed", True), ("bf16", True), ("precise", True), ("bf16", False), ("precise", False)):
        AF.set_mode(mode)
        AF.invalidate_weight_cache()
        decoding.NATIVE_BEAM = native  # True: one library call per step (csrc/decode.hip); False: the python-issued step
        bs._native = None
        for T in (100, 400):
            x, _, _ = synth_batch("video", 1, T, 3, 5049, seed=T, lengths=[T])
            x = x.to(dev)
            t_enc, t_dec, steps = [], [], 0
            for rep in range(args.reps + 1):
                with torch.no_grad():
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    feats = m.proj_encoder(m.frontend(x))
                    enc, _ = m.encoder(feats, None)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    nbest = bs(enc.squeeze(0).float())
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                if rep:  # first repetition = warm-up
                    t_enc.append(t1 - t0)
                    t_dec.append(t2 - t1)
                steps = max(len(h.asdict()["yseq"]) for h in nbest) - 1 if nbest else 0
            enc_ms, dec_ms = min(t_enc) * 1e3, min(t_dec) * 1e3
            assert bool(bs._native) == native
            rows.append({"mode": mode, "step": "native (avsr_beam_step)" if native else "python-issued", "T_frames": T, "beam": args.beam, "encoder_ms": round(enc_ms, 2), "beam_search_ms": round(dec_ms, 2),
                         "longest_hypothesis_tokens": steps, "ms_per_token": round(dec_ms / max(steps, 1), 3),
                         "utterances_per_sec": round(1e3 / (enc_ms + dec_ms), 3)})
            rows[-1]["best_yseq_head"] = nbest[0].asdict()["yseq"][:12] if nbest else []
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    # several utterances in flight: forward_many with 1, 2, 4, 8 workers over `--many` utterances of T = 100 (encoder outputs ready)
    AF.set_mode("mixed")
    AF.invalidate_weight_cache()
    decoding.NATIVE_BEAM = True
    bs._native = None
    encs = []
    with torch.no_grad():
        for i in range(args.many):
            x, _, _ = synth_batch("video", 1, 100, 3, 5049, seed=1000 + i, lengths=[100])
            feats = m.proj_encoder(m.frontend(x.to(dev)))
            encs.append(m.encoder(feats, None)[0].squeeze(0).float())
        ref = None
        for workers in (1, 2, 4, 8):
            best = None
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = bs.forward_many(encs, workers=workers)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            ys = [r[0].asdict()["yseq"] for r in res]
            ref = ys if ref is None else ref
            rows.append({"mode": "mixed", "step": f"native, {workers} searches in flight (forward_many)", "T_frames": 100, "beam": args.beam,
                         "utterances": args.many, "beam_search_ms_total": round(best * 1e3, 1),
                         "utterances_per_sec_search_only": round(args.many / best, 2), "same_best_hypotheses_as_1_worker": ys == ref})
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    AF.set_mode("bf16")
    decoding.NATIVE_BEAM = True
    print(json.dumps({"metric": "decode throughput, video E2E 250M, hybrid CTC/attention beam search (lightning.py:54-64,126-158)",
                      "reference_cpu": "2.12 s per 4 s utterance (T = 100), 8 host cores, BASELINE.md section 2",
                      "data": "synthetic input, synthetic (tests/golden/synth.py) weights", "rows": rows}))


if __name__ == "__main__":
    main()
