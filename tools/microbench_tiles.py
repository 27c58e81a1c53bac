"""GPU sweep of the tuned bf16 kernel's tile / wave-count variants on the front-end convolutions (N = 1600 images)
and a few GEMM shapes, with the XCD-aware tile order on and off, plus the no-MFMA / no-load ablations."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auto_avsr_amd import ops

dev = torch.device("cuda:0")

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

N = 1600
layers = [("l1", 22, 22, 64, 64, 3, 1, 1), ("l2a", 22, 22, 64, 128, 3, 2, 1), ("l2d", 22, 22, 64, 128, 1, 2, 0),
          ("l2", 11, 11, 128, 128, 3, 1, 1), ("l3a", 11, 11, 128, 256, 3, 2, 1), ("l3", 6, 6, 256, 256, 3, 1, 1),
          ("l4a", 6, 6, 256, 512, 3, 2, 1), ("l4", 3, 3, 512, 512, 3, 1, 1)]
rows = []
for name, H, W, Cin, Cout, K, s, p in layers:
    x = torch.randn(N, H, W, Cin, device=dev).bfloat16()
    w = torch.randn(Cout, Cin, K, K, device=dev) / (Cin * K * K) ** 0.5
    wp = ops.conv_weight_permute(w, torch.bfloat16)
    wpd = ops.conv_weight_permute(w, torch.bfloat16, to_dgrad=True)
    OH, OW = ops.conv_out(H, K, s, p), ops.conv_out(W, K, s, p)
    dy = torch.randn(N, OH, OW, Cout, device=dev).bfloat16()
    fl = 2.0 * N * OH * OW * Cout * K * K * Cin
    for op, f in (("fwd", lambda: ops.conv2d_fwd(x, wp, N, H, W, Cin, Cout, K, K, s, p, p, False)),
                  ("dgrad", lambda: ops.conv2d_dgrad(dy, wpd, None, N, H, W, Cin, Cout, K, K, s, p, p, False))):
        res = {}
        for xcd in (1, 0):
            ops.tune(1, xcd)
            for tile in (2, 3, 4, 5, 6, 7, 8):
                ops.tune(0, tile)
                res[f"t{tile}x{xcd}"] = round(timeit(f), 1)
        ops.tune(0, 0); ops.tune(1, 1)
        best = min(res, key=res.get)
        rows.append(dict(layer=name, op=op, gflop=round(fl / 1e9, 1), best=best, best_tflops=round(fl / res[best] / 1e6), **res))
        print(rows[-1], flush=True)
# ablation on the 128x128 forward kernel (l2 / l3 shapes)
for name, H, W, Cin, Cout, K, s, p in [layers[3], layers[5]]:
    x = torch.randn(N, H, W, Cin, device=dev).bfloat16()
    wp = ops.conv_weight_permute(torch.randn(Cout, Cin, K, K, device=dev), torch.bfloat16)
    ops.tune(0, 3)
    r = {}
    for abl in (0, 1, 2):
        ops.tune(2, abl)
        r[f"abl{abl}"] = round(timeit(lambda: ops.conv2d_fwd(x, wp, N, H, W, Cin, Cout, K, K, s, p, p, False)), 1)
    ops.tune(2, 0); ops.tune(0, 0)
    rows.append(dict(layer=name, op="fwd-ablation", **r)); print(rows[-1], flush=True)
# plain NT GEMM
for (M, Nn, K) in [(4096, 4096, 4096), (8192, 8192, 4096), (1600, 3072, 768), (1600, 768, 3072), (1600, 768, 768), (1600, 2304, 768)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(Nn, K, device=dev).bfloat16()
    C = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    res = {}
    for xcd in (1, 0):
        ops.tune(1, xcd)
        for tile in (1, 2, 3, 4, 5, 6, 7, 8):
            res[f"t{tile}x{xcd}"] = round(timeit(lambda: ops.gemm_bf16_nt(A, K, B, K, M, Nn, K, C, Nn, tile=tile)), 1)
    ops.tune(1, 1)
    best = min(res, key=res.get)
    rows.append(dict(gemm=(M, Nn, K), best=best, best_tflops=round(2.0 * M * Nn * K / res[best] / 1e6), **res))
    print(rows[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/microbench_tiles.json", "w"), indent=1)
