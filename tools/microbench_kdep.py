"""How much of a short-K GEMM is prologue/epilogue?  Time vs K at fixed M, N (tuned NT kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for (M, N, tile, cd) in [(1600, 3072, 1, torch.bfloat16), (193600, 128, 2, torch.bfloat16), (193600, 128, 3, torch.bfloat16), (774400, 64, 2, torch.bfloat16), (1600, 768, 1, torch.float32)]:
    for K in (64, 128, 256, 576, 1152, 2304):
        A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=cd)
        us = timeit(lambda: ops.gemm_bf16_nt(A, K, B, K, M, N, K, C, N, tile=tile))
        print(dict(M=M, N=N, K=K, tile=tile, out=str(cd)[6:], us=round(us, 1), tflops=round(2.0*M*N*K/us/1e6, 1)), flush=True)
