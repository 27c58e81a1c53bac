"""GPU: per-shape speed-of-light table of the Conformer-block GEMMs: for the six M = B*T forward contractions of the
training step in the benchmarked arithmetic (f16 activations x two f16 weight planes) -- the launch as the step issues it, the same
tile with its operand stream only (no LDS reads / MFMA) and with LDS reads + MFMA only (no operand loads after the prologue),
against three lower bounds: MFMA (2 planes x 2 M N K at 2.5 PFLOP/s), operand delivery L2 -> LDS (bytes the tile decomposition moves
at the 15 TB/s the feed probe measured, profiles/r2_feed_probe.txt) and the dependent-launch floor (4.4 us: boundary + first tile +
epilogue).  Operands rotate through a pool larger than the Infinity Cache.  -> gpurun_out/microbench_sol.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=40, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


TILES = {1: (64, 64, 31, 32), 7: (128, 64, 33, 34), 5: (256, 128, 35, 36)}  # code -> (BM, BN, feed-only code, mfma-only code)
rows = []
POOL = 24
for name, (M, N, K) in [("attention out / pointwise 2", (1600, 768, 768)), ("FFN w_1", (1600, 3072, 768)), ("FFN w_2", (1600, 768, 3072)),
                        ("fused Q/K/V", (1600, 2304, 768)), ("pointwise 1 (GLU)", (1600, 1536, 768)), ("decoder memory K/V x 6", (1600, 9216, 768))]:
    As = [torch.randn(M, K, device=dev).half() for _ in range(POOL)]
    Ws = [(0.05 * torch.randn(N, 2, K, device=dev)).half() for _ in range(POOL)]
    C = torch.empty(M, N, device=dev, dtype=torch.float16)

    def run(t):
        def f(i):
            W = Ws[i % POOL]
            ops.gemm_h16_nt(As[i % POOL], K, W[:, 0], 2 * K, M, N, K, C, N, tile=t, B_lo=W[:, 1])
        return round(timeit(f), 2)

    t256 = ((M + 255) // 256) * ((N + 127) // 128)
    t12864 = ((M + 127) // 128) * ((N + 63) // 64)
    tile = 5 if t256 >= 160 else (7 if t12864 >= 256 else (2 if (K >= 2048 and t12864 >= 128) else 1))  # gemm_fast.hip's rule
    code = tile if tile in TILES else 7
    BM, BN, c_feed, c_mfma = TILES[code]
    blocks = ((M + BM - 1) // BM) * ((N + BN - 1) // BN)
    feed_bytes = blocks * (BM + 2 * BN) * K * 2.0
    r = dict(gemm=name, shape=(M, N, K), tile=f"{BM}x{BN}" + (" / 3 stages" if tile == 2 else ""), blocks=blocks,
             current_us=run(0), feed_only_us=run(c_feed), mfma_only_us=run(c_mfma),
             bound_mfma_us=round(2 * 2.0 * M * N * K / 2.5e15 * 1e6, 2), bound_feed_us=round(feed_bytes / 15e12 * 1e6, 2),
             bound_hbm_us=round((M * K * 2 + 2 * N * K * 2 + M * N * 4) / 6.3e12 * 1e6, 2), launch_floor_us=4.4)
    r["min_bound_us"] = round(max(r["bound_mfma_us"], r["bound_feed_us"], r["bound_hbm_us"]) + r["launch_floor_us"], 2)
    r["achieved_over_min_bound"] = round(r["current_us"] / r["min_bound_us"], 2)
    r["useful_tflops"] = round(2.0 * M * N * K / r["current_us"] / 1e6)
    rows.append(r)
    print(r, flush=True)
    del As, Ws
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/microbench_sol.json", "w"), indent=1)
