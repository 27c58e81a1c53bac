"""GPU micro-benchmark of the front-end convolutions at the max-frames=1600 shapes (N = 1600 images)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auto_avsr_amd import ops

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20

def timeit(fn, iters=iters, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

N = 1600
layers = [("l1", 22, 22, 64, 64, 3, 1, 1), ("l2a", 22, 22, 64, 128, 3, 2, 1), ("l2", 11, 11, 128, 128, 3, 1, 1),
          ("l3", 6, 6, 256, 256, 3, 1, 1), ("l4", 3, 3, 512, 512, 3, 1, 1)]
rows = []
for name, H, W, Cin, Cout, K, s, p in layers:
    if which not in ("all", name):
        continue
    x = torch.randn(N, H, W, Cin, device=dev).bfloat16()
    w = torch.randn(Cout, Cin, K, K, device=dev) / (Cin * K * K) ** 0.5
    wp = ops.conv_weight_permute(w, torch.bfloat16)
    wpd = ops.conv_weight_permute(w, torch.bfloat16, to_dgrad=True)
    OH, OW = ops.conv_out(H, K, s, p), ops.conv_out(W, K, s, p)
    dy = torch.randn(N, OH, OW, Cout, device=dev).bfloat16()
    fl = 2.0 * N * OH * OW * Cout * K * K * Cin
    for op, f in (("fwd", lambda: ops.conv2d_fwd(x, wp, N, H, W, Cin, Cout, K, K, s, p, p, False)),
                  ("dgrad", lambda: ops.conv2d_dgrad(dy, wpd, None, N, H, W, Cin, Cout, K, K, s, p, p, False)),
                  ("wgrad", lambda: ops.conv2d_wgrad(dy, x, N, H, W, Cin, Cout, K, K, s, p, p, False))):
        us = timeit(f)
        rows.append(dict(layer=name, op=op, us=round(us, 1), tflops=round(fl / us / 1e6, 1)))
        if op == "wgrad":  # output stage of the dedicated 3x3 kernel: partial sums + reduce (default) / atomics / none
            for mode, tag in ((1, "atomics_us"), (2, "no_output_us")):
                ops.tune(3, mode)
                rows[-1][tag] = round(timeit(f), 1)
            ops.tune(3, 0)
            ops.tune(3, 2)
            for abl in (0, 1, 2, 3, 4, 8, 15):
                ops.tune(5, abl); rows[-1][f"abl{abl}"] = round(timeit(f), 1)
            ops.tune(5, 0); ops.tune(3, 0)
        print(rows[-1], flush=True)
if which in ("all", "stem"):
    B, T = 4, 400
    xs = torch.randn(B, T, 88, 88, device=dev)
    w0 = torch.randn(64, 1, 5, 7, 7, device=dev) / 245 ** 0.5
    wp0 = ops.conv_weight_permute(w0, torch.bfloat16, ld_out=248)
    dy0 = torch.randn(B * T, 44, 44, 64, device=dev).bfloat16()
    fl = 2.0 * B * T * 44 * 44 * 64 * 245
    us = timeit(lambda: ops.conv_stem_fwd(xs, wp0, 248, torch.bfloat16, B, T, 88, 88, 64, 5, 7, 7, 2, 2, 3, 3, False))
    rows.append(dict(layer="stem", op="fwd", us=round(us, 1), tflops=round(fl / us / 1e6, 1))); print(rows[-1], flush=True)
    us = timeit(lambda: ops.conv_stem_wgrad(dy0, xs, B, T, 88, 88, 64, 5, 7, 7, 2, 2, 3, 3, False))
    rows.append(dict(layer="stem", op="wgrad", us=round(us, 1), tflops=round(fl / us / 1e6, 1))); print(rows[-1], flush=True)
    us = timeit(lambda: ops.stem357_fwd(xs, w0, B, T, 88, 88))
    rows.append(dict(layer="stem357", op="fwd", us=round(us, 1), tflops=round(fl / us / 1e6, 1))); print(rows[-1], flush=True)
    us = timeit(lambda: ops.stem357_wgrad(dy0, xs, B, T, 88, 88))
    rows.append(dict(layer="stem357", op="wgrad", us=round(us, 1), tflops=round(fl / us / 1e6, 1))); print(rows[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/microbench_conv.json", "w"), indent=1)
