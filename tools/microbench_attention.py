"""Fused attention kernels at the bench shapes (encoder rel-pos self-attention, decoder self / source attention).
GPU box:  python tools/microbench_attention.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def case(tag, B, Tq, Tk, H, relpos, causal, drop):
    D = 64
    g = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
    qu, k, v = g(B, Tq, H, D), g(B, Tk, H, D), g(B, Tk, H, D)
    qv = g(B, Tq, H, D) if relpos else None
    pos = g(2 * Tq - 1, H * D) if relpos else None
    if causal:
        mask = torch.tril(torch.ones(Tq, Tk, dtype=torch.bool, device=dev))[None].expand(B, -1, -1).contiguous()
    else:
        lens = torch.randint(Tk // 2, Tk + 1, (B,), device=dev)
        mask = (torch.arange(Tk, device=dev)[None] < lens[:, None]).unsqueeze(1).contiguous()
    sc = 0.125
    out, lse = ops.attention_fwd(qu, qv, k, v, pos, mask, sc, drop_p=drop, seed=3)
    dout = g(B, Tq, H * D)
    f = timeit(lambda: ops.attention_fwd(qu, qv, k, v, pos, mask, sc, drop_p=drop, seed=3))
    b = timeit(lambda: ops.attention_bwd_dq(qu, qv, k, v, pos, mask, out, lse, dout, sc, drop_p=drop, seed=3))
    a = timeit(lambda: ops.attention_bwd(qu, qv, k, v, pos, mask, out, lse, dout, sc, drop_p=drop, seed=3))
    print(f"{tag:28s} fwd {f:7.1f} us   bwd_dq {b:7.1f} us   bwd(all) {a:7.1f} us")


case("enc relpos B16 T100 H12", 16, 100, 100, 12, True, False, 0.1)
case("enc relpos B4 T400 H12", 4, 400, 400, 12, True, False, 0.1)
case("enc relpos B64 T25 H12", 64, 25, 25, 12, True, False, 0.1)
case("dec self B16 L40 H12", 16, 40, 40, 12, False, True, 0.1)
case("dec src B16 L40 T100 H12", 16, 40, 100, 12, False, False, 0.1)
# round 6: the forward kernel with the key range split over two wave groups (knob 8: 2 = never, 3 = always, 0 = the rule of launch_attn)
for tag, B, Tq, Tk, H, rp in (("dec src B4 L65 T400", 4, 65, 400, 12, False), ("dec src B7 L34 T214", 7, 34, 214, 12, False),
                              ("enc relpos B4 T400", 4, 400, 400, 12, True)):
    for knob in (2, 3, 0):
        ops.tune(8, knob)
        D = 64
        g = lambda *s: (torch.randn(*s, device=dev) * 0.5).half()
        qu, k, v = g(B, Tq, H, D), g(B, Tk, H, D), g(B, Tk, H, D)
        qv = g(B, Tq, H, D) if rp else None
        pos = g(2 * Tq - 1, H * D) if rp else None
        lens = torch.randint(Tk // 2, Tk + 1, (B,), device=dev)
        mask = (torch.arange(Tk, device=dev)[None] < lens[:, None]).unsqueeze(1).contiguous()
        f = timeit(lambda: ops.attention_fwd(qu, qv, k, v, pos, mask, 0.125, drop_p=0.1 if not rp else 0.0, seed=3))
        print(f"{tag:24s} f16 forward, key-split knob {knob}: {f:7.1f} us")
ops.tune(8, 0)
