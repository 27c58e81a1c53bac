"""In-situ-like timing of the skinny GEMMs with cold operands: a pool of distinct (A, W) pairs larger than the 256 MiB
Infinity Cache is cycled, so every launch reads its operands the way the training step does (A written by another kernel,
W not cache-resident).  Sweeps the XCD-aware tile order (avsr_tune knob 1) and the ablations."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
def run(M, N, K, tile, xcd, pool):
    ops.tune(1, xcd)
    n = len(pool)
    for i in range(n): ops.gemm_bf16_nt(pool[i][0], K, pool[i][1], K, M, N, K, pool[i][2], N, tile=tile)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for rep in range(3):
        for i in range(n): ops.gemm_bf16_nt(pool[i][0], K, pool[i][1], K, M, N, K, pool[i][2], N, tile=tile)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * n) * 1e3
rows = []
for (M, N, K) in [(1500, 768, 768), (1500, 768, 3072), (1500, 3072, 768), (1500, 2304, 768), (1500, 1536, 768)]:
    per = (M * K + N * K + M * N) * 2
    n = max(8, int(400e6 // per))
    pool = [(torch.randn(M, K, device=dev).bfloat16(), torch.randn(N, K, device=dev).bfloat16(),
             torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for _ in range(n)]
    res = {}
    for tile in (1, 9, 2, 11, 7):
        for xcd in (2, 1):
            res[f"t{tile}x{xcd}"] = round(run(M, N, K, tile, xcd, pool), 2)
    rows.append(dict(gemm=(M, N, K), pool=n, **res)); print(rows[-1], flush=True)
    del pool
ops.tune(1, 0)
json.dump(rows, open("gpurun_out/microbench_xcd.json", "w"), indent=1)
