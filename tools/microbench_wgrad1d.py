"""GPU: weight gradient of the audio trunk's Conv1d layers (resnet1d.py:38-99: kernel 3 / pad 1 and the 1 x 1 / stride 2 down-sampling
convolutions) at 4 x 16 s of audio (batch A of the audio bench): avsr_conv2d_wgrad_bf16 at several split targets (knob 24).
-> gpurun_out/microbench_wgrad1d.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=4):
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


out = []
N = 4
for (L, Cin, Cout, K, s) in [(64000, 64, 64, 3, 1), (64000, 64, 128, 3, 2), (32000, 128, 128, 3, 1), (32000, 128, 256, 3, 2),
                             (16000, 256, 256, 3, 1), (16000, 256, 512, 3, 2), (8000, 512, 512, 3, 1),
                             (64000, 64, 128, 1, 2), (32000, 128, 256, 1, 2), (16000, 256, 512, 1, 2)]:
    pad = (K - 1) // 2
    OL = (L + 2 * pad - K) // s + 1
    xs = [torch.randn(N, 1, L, Cin, device=dev).bfloat16() for _ in range(3)]
    dys = [torch.randn(N, 1, OL, Cout, device=dev).bfloat16() for _ in range(3)]
    r = dict(L=L, Cin=Cin, Cout=Cout, K=K, stride=s, gflop=round(2e-9 * N * OL * Cout * K * Cin, 2),
             mbytes=round((xs[0].numel() + dys[0].numel()) * 2e-6, 1))
    ref = None
    for target in (0, 128, 256, 512, 2048):
        ops.tune(24, target)
        r[f"t{target or 1024}"] = timeit(lambda i: ops.conv2d_wgrad(dys[i % 3], xs[i % 3], N, 1, L, Cin, Cout, 1, K, s, 0, pad, False))
        g = ops.conv2d_wgrad(dys[0], xs[0], N, 1, L, Cin, Cout, 1, K, s, 0, pad, False)
        if ref is None:
            ref = g
        else:
            r[f"rel{target}"] = float((g - ref).norm() / ref.norm())
    ops.tune(24, 0)
    print(r, flush=True)
    out.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/microbench_wgrad1d.json", "w"), indent=1)
