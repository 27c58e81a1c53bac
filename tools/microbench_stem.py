"""Video stem kernels at the bench geometry (11 x 135 frames of 88 x 88): forward, weight gradient (persistent-grid sweep,
avsr_tune knob 6), and the fused BatchNorm + SiLU + max-pool passes behind them."""
import os, sys, json
sys.path.insert(0, os.getcwd())
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
    return round(s.elapsed_time(e) / iters * 1e3, 1)
B, T, H, W = 11, 135, 88, 88
x = torch.randn(B, T, H, W, device=dev)
w = torch.randn(64, 1, 5, 7, 7, device=dev) * 0.05
res = {"stem_fwd_us": timeit(lambda: ops.stem357_fwd(x, w, B, T, H, W))}
y = ops.stem357_fwd(x, w, B, T, H, W)
dy = torch.randn_like(y)
for g in (512, 768, 1024):
    ops.tune(6, g)
    res[f"stem_wgrad_us_g{g}"] = timeit(lambda: ops.stem357_wgrad(dy, x, B, T, H, W))
ops.tune(6, 0)
C = 64
N, OH, OW = B * T, y.shape[1], y.shape[2]
mean, invstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev) * 0.1
res["bn_act_pool_fwd_us"] = timeit(lambda: ops.bn_act_pool_fwd(y, mean, invstd, gamma, beta, N, OH, OW, C, 3, 2, 1, 1, want_xsel=True))
p, idx, xsel = ops.bn_act_pool_fwd(y, mean, invstd, gamma, beta, N, OH, OW, C, 3, 2, 1, 1, want_xsel=True)
dp = torch.randn_like(p)
sums = ops.bn_bwd_reduce(xsel.view(-1, C), dp.view(-1, C), None, mean, invstd, gamma, beta, dp.numel() // C, C, 1)
res["bn_pool_bwd_reduce_us"] = timeit(lambda: ops.bn_bwd_reduce(xsel.view(-1, C), dp.view(-1, C), None, mean, invstd, gamma, beta, dp.numel() // C, C, 1))
res["bn_pool_bwd_apply_us"] = timeit(lambda: ops.bn_pool_bwd_apply(y, dp, idx, mean, invstd, gamma, beta, sums, 1.0 / (N * OH * OW), N, OH, OW, C, 3, 2, 1, 1))
res["bn_stats_us"] = timeit(lambda: ops.bn_stats_finalize(y.view(-1, C), N * OH * OW, C, 1e-5, 0.1, None, None, None))
print(json.dumps(res), flush=True)
json.dump(res, open("gpurun_out/microbench_stem.json", "w"))
