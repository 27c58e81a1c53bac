"""Do two DIFFERENT small kernels (NT data-gradient GEMM, TN weight-gradient GEMM) overlap on two HIP streams?
Eager launches, long independent chains per stream, one join at the end."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
M, D, F = 1600, 768, 3072
a = torch.randn(M, F, device=dev).bfloat16(); w = torch.randn(D, F, device=dev).bfloat16()
c = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
dy = torch.randn(M, F, device=dev).bfloat16(); x = torch.randn(M, D, device=dev).bfloat16()
dw = torch.empty(F, D, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def nt(): ops.gemm_bf16_nt(a, F, w, F, M, D, F, c, D)
def tn(): ops.gemm_bf16_tn(dy, F, x, D, F, D, M, dw, D)
def run(mode, n=200):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record()
    if mode == "seq":
        for _ in range(n): nt(); tn()
    else:
        s1.wait_stream(cur); s2.wait_stream(cur)
        for _ in range(n):
            with torch.cuda.stream(s1): nt()
            with torch.cuda.stream(s2): tn()
        cur.wait_stream(s1); cur.wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mode in ("seq", "par", "seq", "par"):
    print(mode, round(run(mode), 1), "us per (NT + TN) pair", flush=True)
def only(fn, n=200):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print("nt alone", round(only(nt), 1), "tn alone", round(only(tn), 1))
