"""GPU: the paired data-gradient (NT) + weight-gradient (TN) launch of a Linear backward against its two halves alone, operands
rotating through a pool larger than the Infinity Cache.  Shapes: rows = 1600 tokens; (n_out, n_in) of the encoder's Linear layers."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")
ops.apply_env_tuning()


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


rows, P = 1600, 16
out = []
for n_out, n_in in [(768, 768), (3072, 768), (768, 3072), (2304, 768), (1536, 768)]:
    dys = [torch.randn(rows, n_out, device=dev).bfloat16() for _ in range(P)]
    xs = [torch.randn(rows, n_in, device=dev).bfloat16() for _ in range(P)]
    wTs = [torch.randn(n_in, n_out, device=dev).bfloat16() for _ in range(P)]
    dx = torch.empty(rows, n_in, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(n_out, n_in, device=dev)
    tiles = ((n_out + 63) // 64) * ((n_in + 63) // 64)
    split = 2 if tiles < 300 else 1
    nt = lambda i: ops.gemm_bf16_nt(dys[i % P], n_out, wTs[i % P], n_out, rows, n_in, n_out, dx, n_in)
    tn = lambda i, s=split: ops.gemm_bf16_tn(dys[i % P], n_out, xs[i % P], n_in, n_out, n_in, rows, dw, n_in, accumulate=s > 1, split_k=s)

    def pair(i, s=split):
        with ops.paired():
            tn(i, s)
            nt(i)

    r = dict(n_out=n_out, n_in=n_in, split=split, nt=timeit(nt), tn=timeit(tn), tn_split1=timeit(lambda i: tn(i, 1)),
             tn_split4=timeit(lambda i: tn(i, 4)), pair=timeit(pair), pair_split1=timeit(lambda i: pair(i, 1)))
    r["gflop_each"] = round(2.0 * rows * n_out * n_in / 1e9, 2)
    out.append(r)
    print(r, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/microbench_pair.json", "w"), indent=1)
