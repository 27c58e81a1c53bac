"""GPU: the patch-staged 3x3 convolution kernel (conv_patch.hip, round 6) against the tiled implicit-GEMM kernel it replaces (knob 20 = 1)
on the trunk's stage 2 - 4 shapes at max-frames 1600: f16 forward with two weight planes, bf16 data gradient.  Also a repeat screen:
the patch kernel on the same operands twice must give identical bits.  -> gpurun_out/microbench_patch.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
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


rows = []
NB = 1600
bad = 0
for name, H, Cin, Cout in [("l2", 11, 128, 128), ("l3", 6, 256, 256), ("l4", 3, 512, 512)]:
    xs = [torch.randn(NB, H, H, Cin, device=dev).half() for _ in range(4)]
    ws = [(0.05 * torch.randn(Cout, 2, 9 * Cin, device=dev)).half() for _ in range(4)]
    dys = [torch.randn(NB, H, H, Cout, device=dev).bfloat16() for _ in range(4)]
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    wpd = ops.conv_weight_permute(w, torch.bfloat16, to_dgrad=True)
    fl = 2.0 * NB * H * H * Cout * 9 * Cin
    res = {}
    for knob, tag in ((1, "tiled"), (0, "patch"), (3, "patch3")):
        ops.tune(20, knob)
        res[f"fwd2_{tag}"] = timeit(lambda i: ops.conv2d_fwd(xs[i % 4], ws[i % 4], NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=2))
        res[f"fwd1_{tag}"] = timeit(lambda i: ops.conv2d_fwd(xs[i % 4], ws[i % 4], NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=1))
        res[f"dgrad_{tag}"] = timeit(lambda i: ops.conv2d_dgrad(dys[i % 4], wpd, None, NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False))
    ops.tune(20, 0)
    # repeat screen + agreement with the tiled kernel
    for r in range(20):
        a = ops.conv2d_fwd(xs[r % 4], ws[r % 4], NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=2)
        b = ops.conv2d_fwd(xs[r % 4], ws[r % 4], NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=2)
        d1 = ops.conv2d_dgrad(dys[r % 4], wpd, None, NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False)
        d2 = ops.conv2d_dgrad(dys[r % 4], wpd, None, NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False)
        if not (torch.equal(a, b) and torch.equal(d1, d2)):
            bad += 1
            print(f"REPEAT MISMATCH {name} rep {r}", flush=True)
    ops.tune(20, 1)
    at = ops.conv2d_fwd(xs[0], ws[0], NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=2)
    dt = ops.conv2d_dgrad(dys[0], wpd, None, NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False)
    ops.tune(20, 0)
    ap = ops.conv2d_fwd(xs[0], ws[0], NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False, wp_planes=2)
    dp = ops.conv2d_dgrad(dys[0], wpd, None, NB, H, H, Cin, Cout, 3, 3, 1, 1, 1, False)
    res["fwd_rel_vs_tiled"] = float((ap.float() - at.float()).norm() / at.float().norm())
    res["dgrad_rel_vs_tiled"] = float((dp.float() - dt.float()).norm() / dt.float().norm())
    rows.append(dict(conv=name, gflop=round(fl / 1e9, 1), **res,
                     fwd2_tflops_useful=round(fl / res["fwd2_patch"] / 1e6), dgrad_tflops=round(fl / res["dgrad_patch"] / 1e6)))
    print(rows[-1], flush=True)
print("REPEAT SCREEN", "FAILED" if bad else "clean")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/microbench_patch.json", "w"), indent=1)
