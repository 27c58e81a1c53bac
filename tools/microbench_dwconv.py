import sys, os
sys.path.insert(0, os.getcwd())
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, T, C, K = 4, 400, 768, 31
x = torch.randn(B, T, C, device=dev).bfloat16(); dy = torch.randn(B, T, C, device=dev).bfloat16()
dw = torch.zeros(C, K, device=dev); db = torch.zeros(C, device=dev)
print("dwconv_wgrad us", round(timeit(lambda: ops.call("avsr_dwconv_wgrad", x.data_ptr(), dy.data_ptr(), 1, dw.data_ptr(), db.data_ptr(), B, T, C, K, ops._stream(x))), 1))
