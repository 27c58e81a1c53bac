"""Ring-depth sweep on the skinny M = B*T GEMMs of the transformer layers (one 64x64 / 128x64 block per CU: the bytes a CU
keeps in flight are the ring depth).  tile codes: gemm_fast.hip launch_tile."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
ops.apply_env_tuning()
def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
rows = []
for (M, N, K) in [(1600, 768, 768), (1600, 768, 3072), (1600, 1536, 768), (1600, 3072, 768), (1600, 2304, 768),
                  (800, 768, 768), (1600, 5056, 768), (260, 768, 768), (260, 5056, 768)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = A.float() @ B.float().t()
    res = {}
    for tile in [int(t) for t in os.environ.get('TILES', '1,9,10,2,7,11,12,4,13').split(',')]:
        res[f"t{tile}"] = round(timeit(lambda: ops.gemm_bf16_nt(A, K, B, K, M, N, K, C, N, tile=tile)), 2)
        err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
        assert err < 2e-2 or os.environ.get('NOCHECK'), (tile, err)
    best = min(res, key=res.get)
    rows.append(dict(gemm=(M, N, K), best=best, tflops=round(2.0 * M * N * K / res[best] / 1e6), **res))
    print(rows[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/microbench_ring.json", "w"), indent=1)
