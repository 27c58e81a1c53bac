"""Do independent small GEMMs overlap when issued on separate HIP streams (inside one captured graph)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
M, N, K = 1600, 768, 768
As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(4)]
Bs = [torch.randn(N, K, device=dev).bfloat16() for _ in range(4)]
Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]

def seq(n):
    for i in range(n):
        ops.gemm_bf16_nt(As[i], K, Bs[i], K, M, N, K, Cs[i], N, tile=1)

def par(n):
    cur = torch.cuda.current_stream()
    for i in range(n):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            ops.gemm_bf16_nt(As[i], K, Bs[i], K, M, N, K, Cs[i], N, tile=1)
    for i in range(n):
        cur.wait_stream(streams[i])

def bench(fn, n, reps=20):
    # capture `reps` repetitions in a graph so that launch overhead is off the table
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(n)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn(n)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 / reps * 1e3

for n in (1, 2, 3, 4):
    print(f"n={n}: sequential {bench(seq, n):.1f} us   parallel streams {bench(par, n):.1f} us", flush=True)
