"""GPU: the element-wise middle of the Conformer ConvolutionModule (GLU -> depthwise conv -> BatchNorm -> SiLU) at the benchmarked
shape (B*T = 1600 frames, C = 768, K = 31): the launches of rounds 2 - 5 (dwconv | bn_small_fwd forward; bn_small_bwd | dwconv_wgrad |
dwconv(flip + GLU backward) backward) against the fused launches of round 6 (csrc/convmod_fused.hip), in the mixed mode's dtypes
(f32 chain forward -> f16 result + bf16 twins; bf16 backward) and the bf16 mode's.  Back-to-back launches on one stream, 4 operand
sets in rotation.  -> gpurun_out/microbench_convmod.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=50, warm=5):
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
for (B, T) in [(16, 100), (8, 200), (4, 400), (32, 50), (3, 512)]:
    C, K = 768, 31
    rows = B * T
    for tag, adt, sdt in (("mixed", torch.float32, torch.float16), ("bf16", torch.bfloat16, torch.bfloat16)):
        a = [(torch.randn(rows, 2 * C, device=dev) * 1.2).to(adt) for _ in range(4)]
        ab = [x.bfloat16() for x in a]
        w = torch.randn(C, K, device=dev) * 0.2
        b = torch.randn(C, device=dev) * 0.1
        g, bt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        od = sdt if sdt != adt else None
        c0 = ops.dwconv(a[0], w, b, B, T, C, K, glu_in=True).view(rows, C)
        s0, mean, inv = ops.bn_small_fwd(c0, rows, C, g, bt, 1e-5, 0.1, rm, rv, nbt, 1, out_dtype=od)
        cb = c0.bfloat16()
        ds = [torch.randn(rows, C, device=dev).bfloat16() for _ in range(4)]
        dw, db = torch.zeros(C, K, device=dev), torch.zeros(C, device=dev)
        r = dict(B=B, T=T, mode=tag)
        r["fwd_dwconv"] = timeit(lambda i: ops.dwconv(a[i % 4], w, b, B, T, C, K, glu_in=True))
        r["fwd_bn_small"] = timeit(lambda i: ops.bn_small_fwd(c0, rows, C, g, bt, 1e-5, 0.1, rm, rv, nbt, 1, out_dtype=od))

        def sep_fwd(i):
            c = ops.dwconv(a[i % 4], w, b, B, T, C, K, glu_in=True).view(rows, C)
            ops.bn_small_fwd(c, rows, C, g, bt, 1e-5, 0.1, rm, rv, nbt, 1, out_dtype=od)

        r["fwd_separate"] = timeit(sep_fwd)
        r["fwd_fused"] = timeit(lambda i: ops.convmod_dwbn_fwd(a[i % 4], w, b, B, T, C, K, g, bt, 1e-5, 0.1, rm, rv, nbt, out_dtype=od))
        r["bwd_bn_small"] = timeit(lambda i: ops.bn_small_bwd(cb, ds[i % 4], rows, C, mean, inv, g, bt, 1))
        r["bwd_wgrad"] = timeit(lambda i: ops.dwconv_wgrad(ab[i % 4], ds[i % 4], dw, db, B, T, C, K, glu_in=True))
        r["bwd_dgrad_glu"] = timeit(lambda i: ops.dwconv(ds[i % 4], w, None, B, T, C, K, flip=True, glu_a=ab[i % 4]))

        def sep_bwd(i):
            dc, _, _ = ops.bn_small_bwd(cb, ds[i % 4], rows, C, mean, inv, g, bt, 1)
            ops.dwconv_wgrad(ab[i % 4], dc, dw, db, B, T, C, K, glu_in=True)
            ops.dwconv(dc, w, None, B, T, C, K, flip=True, glu_a=ab[i % 4])

        r["bwd_separate"] = timeit(sep_bwd)
        r["bwd_fused"] = timeit(lambda i: ops.convmod_dwbn_bwd(ab[i % 4], cb, ds[i % 4], mean, inv, g, bt, w, B, T, C, K, dw, db))
        print(r, flush=True)
        out.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/microbench_convmod.json", "w"), indent=1)
