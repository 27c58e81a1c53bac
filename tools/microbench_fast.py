"""GPU micro-benchmark: tuned LDS-DMA NT kernel (avsr_gemm_bf16_nt) vs the generic register-staged kernel."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auto_avsr_amd import ops

dev = torch.device("cuda:0")

def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

rows = []
shapes = [(1600, 3072, 768), (1600, 768, 3072), (1600, 1536, 768), (1600, 768, 768), (768, 768, 1600), (3072, 768, 1600), (260, 768, 768), (260, 3072, 768)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    ref = None
    if M * N <= 4096 * 4096:
        ref = A.float() @ B.float().t()
    for tile in (1, 2, 3):
        for split in (1,):
            if split > 1 and (K // 64) < 2 * split:
                continue
            if split > 1 and M * N > 1600 * 3072:
                continue
            C = torch.zeros(M, N, device=dev, dtype=torch.float32 if split > 1 else torch.bfloat16)
            f = lambda: ops.gemm_bf16_nt(A, K, B, K, M, N, K, C, N, tile=tile, accumulate=split > 1, split_k=split)
            us = timeit(f, iters=30 if M > 100000 else 100)
            err = -1.0
            if ref is not None and split == 1:
                err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
            r = dict(M=M, N=N, K=K, kernel="fast", tile=tile, split=split, us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1), rel_err=round(err, 5))
            rows.append(r); print(r, flush=True)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for tile in (64, 128):
        f = lambda: ops.gemm(0, A, K, B, K, M, N, K, C, N, force_tile=tile)
        us = timeit(f, iters=30 if M > 100000 else 100)
        r = dict(M=M, N=N, K=K, kernel="generic", tile=tile, us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1))
        rows.append(r); print(r, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/microbench_fast.json", "w"), indent=1)
