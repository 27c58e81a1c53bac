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
for (M, N, K) in [(768, 768, 1600), (3072, 768, 1600), (768, 3072, 1600), (1536, 768, 1600), (5056, 768, 1600)]:
    A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
    ref = A.float().t() @ B.float()
    for split in (1, 2, 4):
        C = torch.zeros(M, N, device=dev)
        us = timeit(lambda: ops.gemm_bf16_tn(A, M, B, N, M, N, K, C, N, accumulate=split > 1, split_k=split))
        C.zero_(); ops.gemm_bf16_tn(A, M, B, N, M, N, K, C, N, accumulate=split > 1, split_k=split)
        err = ((C - ref).abs().max() / ref.abs().max()).item()
        print(dict(M=M, N=N, K=K, kernel="tn_tr", split=split, us=round(us, 1), tflops=round(2.0*M*N*K/us/1e6, 1), err=round(err, 5)), flush=True)
    def viaT():
        At = ops.transpose_cast(A, K, M); Bt = ops.transpose_cast(B, K, N)
        Cc = torch.empty(M, N, device=dev)
        ops.gemm_bf16_nt(At, At.shape[1], Bt, Bt.shape[1], M, N, At.shape[1], Cc, N, tile=1)
    print(dict(M=M, N=N, K=K, kernel="transpose+nt", us=round(timeit(viaT), 1)), flush=True)
