"""GPU ablation (round 5; round 6: tiles 27 - 30 = 256-row tiles on 8 waves): the split-plane convolution of ResNet stages 1 - 2 with the A tile converted to split8 in LDS (tiles 13 /
14: what the mixed mode runs) against the same kernel on an A operand that ARRIVES pre-split (tiles 23 - 26: no conversion pass,
one barrier per k-tile).  Timing only (the f32 input is read as if it were split8).  -> gpurun_out/microbench_presplit.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=2):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


rows = []
N = 1600
for name, H, W, Cin, Cout, K, s, p in [("l1", 22, 22, 64, 64, 3, 1, 1), ("l2a", 22, 22, 64, 128, 3, 2, 1), ("l2d", 22, 22, 64, 128, 1, 2, 0),
                                       ("l2", 11, 11, 128, 128, 3, 1, 1)]:
    xs = [torch.randn(N, H, W, Cin, device=dev) for _ in range(3)]
    w = torch.randn(Cout, Cin, K, K, device=dev) / (Cin * K * K) ** 0.5
    wp = ops.conv_weight_permute_split(w)
    OH = ops.conv_out(H, K, s, p)
    y = torch.empty(N, OH, OH, Cout, device=dev)
    y2 = torch.empty(N, OH, OH, Cout, device=dev, dtype=torch.bfloat16)
    st = torch.empty(ops.bn_stat_tiles(N * OH * OH), 2, Cout, device=dev)
    res = {}
    for tile in ((13, 23, 25, 27, 28) if Cout < 128 else (14, 24, 26, 13, 23, 27, 29, 30)):
        def f(i, tile=tile):
            ops.call("avsr_conv2d_f32s_stats", ops._ptr(xs[i % 3]), ops._ptr(wp), ops._ptr(y), ops._ptr(ops.zero_page(dev)), N, H, W, Cin,
                     Cout, K, K, s, p, p, tile, 1, ops._ptr(y2), ops._ptr(st), st.shape[0], ops._stream(y))
        res[f"t{tile}"] = round(timeit(f), 1)
    rows.append(dict(conv=name, gflop=round(2.0 * N * OH * OH * Cout * K * K * Cin / 1e9, 1), **res))
    print(rows[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/microbench_presplit.json", "w"), indent=1)
