"""conv3x3_c64 kernel alone at the trunk's first-stage geometry (1600 x 22 x 22 x 64), with its ablation knob (avsr_tune 13)
and against the tiled kernel (knob 12 = 1).  GPU box: python tools/microbench_c64.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")
N, H, W, C = 1600, 22, 22, 64
x = torch.randn(N, H, W, C, device=dev).bfloat16()
res = torch.randn(N, H, W, C, device=dev).bfloat16()
w = torch.randn(C, C, 3, 3, device=dev) / 24
wp = ops.conv_weight_permute(w, torch.bfloat16)
wpd = ops.conv_weight_permute(w, torch.bfloat16, to_dgrad=True)


def t(fn, reps=20):
    """us per call, replayed from a hipGraph (the Python side of one call costs about as much as the kernel)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


fwd = lambda: ops.conv2d_fwd(x, wp, N, H, W, C, C, 3, 3, 1, 1, 1, False)
dgr = lambda: ops.conv2d_dgrad(x, wpd, res, N, H, W, C, C, 3, 3, 1, 1, 1, False)
ops.tune(12, 1)
print(f"tiled kernel      : fwd {t(fwd):7.1f} us   dgrad+resid {t(dgr):7.1f} us")
ops.tune(12, 0)
for abl, name in ((0, "full"), (2, "no staging"), (4, "no copy-out"), (6, "no staging, no copy-out"),
                  (8 + 6, "MFMA loop without fragment reads, no staging, no copy-out"),
                  (16 + 6, "fragment reads without MFMAs, no staging, no copy-out")):
    ops.tune(13, abl)
    print(f"persistent {name:60s}: fwd {t(fwd):7.1f} us   dgrad+resid {t(dgr):7.1f} us")
ops.tune(13, 0)
