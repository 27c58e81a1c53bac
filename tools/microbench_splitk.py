"""GPU probe (round 6): k split of the long-K, small-output forward GEMMs (FFN w_2: 1600 x 768 x 3072, two f16 weight planes) with
f32 atomics onto a zeroed output -- 156 blocks of 48 k-tiles become 312 of 24.  Knob 26 of avsr_gemm_h16_nt."""
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
    return round(s.elapsed_time(e) / iters * 1e3, 1)


for (M, N, K) in [(1600, 768, 3072), (1600, 768, 768), (1600, 768, 2304), (640, 768, 3072)]:
    A = [torch.randn(M, K, device=dev).half() for _ in range(4)]
    W = [(0.03 * torch.randn(N, 2, K, device=dev)).half() for _ in range(4)]
    bias = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev)
    C = [torch.zeros(M, N, device=dev) for _ in range(4)]
    row = dict(M=M, N=N, K=K)
    for tile in (0, 1, 2, 7):
        for sk in (1, 2, 3, 4):
            ops.tune(26, sk)
            row[f"tile{tile}_split{sk}"] = timeit(lambda i: ops.gemm_h16_nt(A[i % 4], K, W[i % 4][:, 0], 2 * K, M, N, K, C[i % 4], N, bias=bias, alpha=0.5,
                                                                            resid=resid, ldr=N, tile=tile, B_lo=W[i % 4][:, 1]))
    ops.tune(26, 0)
    print(row, flush=True)
