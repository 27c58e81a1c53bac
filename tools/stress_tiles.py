"""GPU race screen of the round-6 8-wave tiles: the 256x128 two-plane GEMM / convolution tiles and the 256x64 split-plane convolution
tile against the 4-wave tiles they replace, on the shapes of the training step, many repetitions on fresh data, other work in
flight.  Every output element sums the same products in the same order in both tiles, so the results must be BIT-identical (f16
result AND bf16 twin); any difference is a synchronisation bug."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_avsr_amd import functional as AF
from auto_avsr_amd import ops

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
twins = []
ops.TWIN = lambda y: twins.append(torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)) or twins[-1]
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device=dev)
for (M, N, K) in [(1600, 3072, 768), (1595, 3072, 768), (1600, 9216, 768)]:
    for r in range(reps):
        A = torch.randn(M, K, device=dev).half()
        W = (0.05 * torch.randn(N, 2, K, device=dev)).half()
        outs = []
        with torch.cuda.stream(side):  # unrelated traffic beside the kernels under test
            noise_a.mul_(1.0001)
        for t in (7, 5):
            C = torch.empty(M, N, device=dev, dtype=torch.float16)
            C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ops.call("avsr_gemm_h16_nt", ops._ptr(A), K, ops._ptr(W[:, 0]), ops._ptr(W[:, 1]), 2 * K, M, N, K, None, 1, 0.0, 0, None, 1.0, None,
                     0, 0, ops._ptr(C), 2, N, t, ops._ptr(C2), N, ops._stream(A))
            outs.append((C, C2))
        if not (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])):
            bad += 1
            d = (outs[0][0].float() - outs[1][0].float()).abs()
            print(f"MISMATCH gemm {(M, N, K)} rep {r}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}", flush=True)
    print(f"gemm {(M, N, K)}: {reps} repetitions compared", flush=True)
NB = 1600
for name, H, W_, Cin, Cout, Kk, s, p in [("l3", 6, 6, 256, 256, 3, 1, 1), ("l4", 3, 3, 512, 512, 3, 1, 1), ("l3a", 11, 11, 128, 256, 3, 2, 1)]:
    for r in range(reps // 3):
        x = torch.randn(NB, H, W_, Cin, device=dev).half()
        w = (0.05 * torch.randn(Cout, 2, Kk * Kk * Cin, device=dev)).half()
        outs = []
        for t in (7, 5):
            ops.tune(18, t)
            twins.clear()
            y = ops.conv2d_fwd(x, w, NB, H, W_, Cin, Cout, Kk, Kk, s, p, p, False, wp_planes=2)
            outs.append((y, twins[-1]))
        ops.tune(18, 0)
        if not (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])):
            bad += 1
            print(f"MISMATCH conv {name} rep {r}", flush=True)
    print(f"conv {name}: {reps // 3} repetitions compared", flush=True)
torch.cuda.synchronize()
print("RACE SCREEN", "FAILED" if bad else "clean", bad)
sys.exit(1 if bad else 0)
