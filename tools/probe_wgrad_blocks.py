import os, sys
sys.path.insert(0, os.getcwd())
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3
N = 1600
for (H, C) in ((22, 64), (11, 128), (6, 256), (3, 512)):
    xs = [torch.randn(N, H, H, C, device=dev).bfloat16() for _ in range(3)]
    dys = [torch.randn(N, H, H, C, device=dev).bfloat16() for _ in range(3)]
    i = [0]
    def fn():
        i[0] += 1
        ops.conv2d_wgrad(dys[i[0] % 3], xs[i[0] % 3], N, H, H, C, C, 3, 3, 1, 1, 1, False, torch_layout=True)
    line = f"{H:2d}x{H:<2d} C={C:3d}: main+reduce us at block targets"
    for target in (0, 256, 320, 384, 448, 512, 640, 768, 1024):
        ops.tune(15, target)
        line += f"  {target or 'auto'}:{t(fn):6.1f}"
    ops.tune(15, 0)
    print(line, flush=True)
