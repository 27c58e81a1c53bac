import os, sys
sys.path.insert(0, os.getcwd())
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3
D = 768
for rows in (100, 400, 1600, 3200):
    x = torch.randn(rows, D, device=dev); gamma = torch.randn(D, device=dev); beta = torch.randn(D, device=dev)
    mean, rstd = x.mean(1), 1.0 / x.std(1)
    dres = torch.randn(rows, D, device=dev)
    dgamma, dbeta, gsum = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    gout = torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
    dy = torch.randn(rows, D, device=dev).bfloat16()
    full = t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres, gout=gout, gsum=gsum, alpha=0.5, drop_p=0.1, seed=3))
    plain = t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=dres))
    fwd = t(lambda: ops.layernorm_fwd(x, gamma, beta, torch.bfloat16))
    sd = t(lambda: ops.scale_dropout(x, torch.bfloat16))
    print(f"rows {rows}: bwd full {full:5.1f}  bwd plain {plain:5.1f}  fwd {fwd:5.1f}  cast {sd:5.1f}")
