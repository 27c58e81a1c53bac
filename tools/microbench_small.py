"""Small [B*T, D] kernels of a Conformer layer alone (graph-replayed, us per launch): LayerNorm backward with and without its
atomics-carrying outputs, the depthwise-convolution weight gradient, the small BatchNorm pair.  GPU box: python tools/microbench_small.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")


def t(fn, reps=20):
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
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


rows, D = 1600, 768
x = torch.randn(rows, D, device=dev)
gamma = torch.randn(D, device=dev)
mean, rstd = x.mean(1), 1.0 / x.std(1)
dres = torch.randn(rows, D, device=dev)
dgamma, dbeta, gsum = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
gout = torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
for name, dy in (("bf16 dy", torch.randn(rows, D, device=dev).bfloat16()), ("f32 dy", torch.randn(rows, D, device=dev))):
    full = t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres, gout=gout, gsum=gsum, alpha=0.5, drop_p=0.1, seed=3))
    nogs = t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres, gout=gout, alpha=0.5, drop_p=0.1, seed=3))
    plain = t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres))
    print(f"layernorm_bwd {name}: with gout + gsum {full:6.1f}   with gout {nogs:6.1f}   dx + dgamma/dbeta only {plain:6.1f}")
    for rpw in (1, 2, 3, 4, 8):  # knob 25: rows per wave of the dx blocks
        ops.tune(25, rpw)
        full = t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres, gout=gout, gsum=gsum, alpha=0.5, drop_p=0.1, seed=3))
        nogs = t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres, gout=gout, alpha=0.5, drop_p=0.1, seed=3))
        print(f"   rows per wave {rpw}: with gout + gsum {full:6.1f}   with gout {nogs:6.1f}")
    ops.tune(25, 0)

B, T, K = 4, 400, 31
a = torch.randn(B, T, 2 * D, device=dev).bfloat16()
dyc = torch.randn(B, T, D, device=dev).bfloat16()
dw, db = torch.zeros(D, K, device=dev), torch.zeros(D, device=dev)
print(f"dwconv_wgrad (GLU folded in, bf16, 4 x 400 x 768): {t(lambda: ops.dwconv_wgrad(a, dyc, dw, db, B, T, D, K, glu_in=True)):6.1f}")
B, T = 16, 100
a = torch.randn(B, T, 2 * D, device=dev).bfloat16()
dyc = torch.randn(B, T, D, device=dev).bfloat16()
print(f"dwconv_wgrad (GLU folded in, bf16, 16 x 100 x 768): {t(lambda: ops.dwconv_wgrad(a, dyc, dw, db, B, T, D, K, glu_in=True)):6.1f}")
