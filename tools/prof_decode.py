"""One beam search (video E2E 250M, synthetic weights, beam 40) for `rocprofv3 --kernel-trace`: the encoder runs in the bf16
mode so that every split-plane GEMM launch in the trace belongs to the decoding steps.  python tools/prof_decode.py [T] [native]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    native = (sys.argv[2] if len(sys.argv) > 2 else "1") != "0"
    import lightning
    from synth import synth_batch, synth_state_dict

    from auto_avsr_amd import decoding
    from auto_avsr_amd import functional as AF
    from auto_avsr_amd.e2e import E2E

    decoding.NATIVE_BEAM = native
    from auto_avsr_amd import ops

    ops.apply_env_tuning()  # AVSR_TUNE="17=1": the LDS-staged linear-layer kernel of the decoding step
    dev = torch.device("cuda:0")
    m = E2E(5049, "video")
    m.load_state_dict(synth_state_dict(m.state_dict(), 3))
    m = m.to(dev).eval()
    bs = lightning.get_beam_search_decoder(m, [str(i) for i in range(5049)], beam_size=40)
    AF.set_mode("bf16")
    x, _, _ = synth_batch("video", 1, T, 3, 5049, seed=T, lengths=[T])
    with torch.no_grad():
        feats = m.proj_encoder(m.frontend(x.to(dev)))
        enc, _ = m.encoder(feats, None)
        e = enc.squeeze(0).float()
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nbest = bs(e)
            torch.cuda.synchronize()
            print(f"search {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms, {len(nbest)} hypotheses, longest {max(len(h.yseq) for h in nbest)}")


if __name__ == "__main__":
    main()
