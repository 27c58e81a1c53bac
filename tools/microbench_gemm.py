"""GPU micro-benchmark of the hand-written MFMA GEMM family on the hot-path shapes
(M = 1600 tokens; D=768, F=3072).  hipBLASLt (torch.matmul) is timed beside it as a yardstick only."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auto_avsr_amd import ops

dev = torch.device("cuda:0")

def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us

rows = []
shapes = [(1600, 3072, 768), (1600, 768, 3072), (1600, 2304, 768), (1600, 768, 768), (1600, 5049, 768),
          (4096, 4096, 4096)]
for (M, N, K) in shapes:
    for layout, name in ((0, "NT fwd"), (1, "NN dgrad"), (2, "TN wgrad")):
        for tile in (64, 128):
            A = torch.randn(M, K, device=dev).bfloat16()
            B = torch.randn(N, K, device=dev).bfloat16()
            As = A if layout != 2 else A.t().contiguous()
            Bs = B if layout == 0 else B.t().contiguous()
            lda = As.shape[1]; ldb = Bs.shape[1]
            if lda % 8 or ldb % 8:
                continue
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            f = lambda: ops.gemm(layout, As, lda, Bs, ldb, M, N, K, C, N, force_tile=tile)
            us = timeit(f)
            ref = (A.float() @ B.float().t())
            err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
            tf = 2.0 * M * N * K / us / 1e6
            g = lambda: torch.matmul(A, B.t())
            us_ref = timeit(g)
            rows.append(dict(M=M, N=N, K=K, layout=name, tile=tile, us=round(us, 1), tflops=round(tf, 1),
                             hipblaslt_us=round(us_ref, 1), rel_err=err))
            print(rows[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/microbench_gemm.json", "w"), indent=1)
