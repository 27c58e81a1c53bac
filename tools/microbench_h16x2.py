"""GPU: the f16 forward GEMM / convolution with ONE and with TWO weight planes (round 5, gemm_fast_kernel.h WP = 2), per tile
code, on the shapes of the training step.  Operands rotate through a pool larger than the Infinity Cache (the step's operands
are not cache-resident either).  -> gpurun_out/microbench_h16x2.json"""
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


rows = []
POOL = 24
for (M, N, K) in [(1600, 768, 768), (1600, 3072, 768), (1600, 768, 3072), (1600, 2304, 768), (1600, 1536, 768), (1600, 9216, 768),
                  (260, 768, 768), (260, 2048, 768)]:
    As = [torch.randn(M, K, device=dev).half() for _ in range(POOL)]
    Ws = [(0.05 * torch.randn(N, 2, K, device=dev)).half() for _ in range(POOL)]
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    res = {}
    for planes, tiles in ((1, (0, 1, 7, 4)), (2, (0, 1, 21, 7, 2, 4, 5, 8, 22, 23))):
        for t in tiles:
            def f(i, t=t, planes=planes):
                W = Ws[i % POOL]
                ops.gemm_h16_nt(As[i % POOL], K, W[:, 0], 2 * K, M, N, K, C, N, tile=t, B_lo=W[:, 1] if planes == 2 else None)
            res[f"p{planes}t{t}"] = round(timeit(f), 2)
    rows.append(dict(gemm=(M, N, K), tflops_p1=round(2.0 * M * N * K / res["p1t0"] / 1e6), tflops_p2=round(2.0 * M * N * K / res["p2t0"] / 1e6), **res))
    print(rows[-1], flush=True)
    del As, Ws

NB = 1600
for name, H, W_, Cin, Cout, Kk, s, p in [("l3a", 11, 11, 128, 256, 3, 2, 1), ("l3d", 11, 11, 128, 256, 1, 2, 0), ("l3", 6, 6, 256, 256, 3, 1, 1),
                                         ("l4a", 6, 6, 256, 512, 3, 2, 1), ("l4", 3, 3, 512, 512, 3, 1, 1)]:
    xs = [torch.randn(NB, H, W_, Cin, device=dev).half() for _ in range(4)]
    ws = [(0.05 * torch.randn(Cout, 2, Kk * Kk * Cin, device=dev)).half() for _ in range(4)]
    OH = ops.conv_out(H, Kk, s, p)
    fl = 2.0 * NB * OH * OH * Cout * Kk * Kk * Cin
    res = {}
    for planes, tiles in ((1, (0,)), (2, (7, 2, 4, 5, 8, 22, 23))):
        for t in tiles:
            ops.tune(18, t)
            res[f"p{planes}t{t}"] = round(timeit(lambda i: ops.conv2d_fwd(xs[i % 4], ws[i % 4], NB, H, W_, Cin, Cout, Kk, Kk, s, p, p, False,
                                                                          wp_planes=planes), iters=20), 1)
    ops.tune(18, 0)
    rows.append(dict(conv=name, gflop=round(fl / 1e9, 1), **res))
    print(rows[-1], flush=True)
    del xs, ws
# bf16 data gradients of the same layers on the one-plane tiles (knob 0): 4 = 128x128 / 2 stages (default), 5 = 256x128 / 8 waves,
# 3 = 128x128 / 3 stages, 8 = 256x64 / 8 waves
for name, H, W_, Cin, Cout, Kk, s, p in [("l2", 11, 11, 128, 128, 3, 1, 1), ("l3", 6, 6, 256, 256, 3, 1, 1), ("l4", 3, 3, 512, 512, 3, 1, 1)]:
    OH = ops.conv_out(H, Kk, s, p)
    dys = [torch.randn(NB, OH, OH, Cout, device=dev).bfloat16() for _ in range(4)]
    w = torch.randn(Cout, Cin, Kk, Kk, device=dev) / (Cin * Kk * Kk) ** 0.5
    wpd = ops.conv_weight_permute(w, torch.bfloat16, to_dgrad=True)
    res = {}
    for t in (4, 5, 3, 8):
        ops.tune(0, t)
        res[f"dgrad_t{t}"] = round(timeit(lambda i: ops.conv2d_dgrad(dys[i % 4], wpd, None, NB, H, W_, Cin, Cout, Kk, Kk, s, p, p, False), iters=20), 1)
    ops.tune(0, 0)
    rows.append(dict(conv=name, **res))
    print(rows[-1], flush=True)
    del dys
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/microbench_h16x2.json", "w"), indent=1)
