"""Tile sweep on the M = B*T GEMM shapes of the transformer layers (2-wave 64x64 and 64x128 variants were tried and
measured 0-20 % slower than the 4-wave 64x64 / 128x64 tiles; the M = 800 row shows the ~7 us latency floor at K = 768)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, K) in [(1600, 768, 768), (1600, 768, 3072), (1600, 1536, 768), (1600, 3072, 768), (1600, 2304, 768), (800, 768, 768)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = A.float() @ B.float().t()
    res = {}
    for tile in (1, 2, 4, 7):
        res[f"t{tile}"] = round(timeit(lambda: ops.gemm_bf16_nt(A, K, B, K, M, N, K, C, N, tile=tile)), 1)
        err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
        assert err < 2e-2, (tile, err)
    print((M, N, K), res, flush=True)
