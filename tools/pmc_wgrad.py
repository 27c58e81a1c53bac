"""Workload for rocprofv3 passes: the dedicated 3x3 weight-gradient kernel at two trunk geometries, ablation modes in
order (tools/rocpd_list.py prints the per-dispatch durations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from auto_avsr_amd import ops
dev = torch.device("cuda:0")
N = 1600
modes = [int(a) for a in sys.argv[1:]] or [0]
for (H, C) in [(22, 64), (11, 128), (3, 512)]:
    x = torch.randn(N, H, H, C, device=dev).bfloat16()
    dy = torch.randn(N, H, H, C, device=dev).bfloat16()
    for abl in modes:
        ops.tune(5, abl)
        for _ in range(3):
            ops.conv2d_wgrad(dy, x, N, H, H, C, C, 3, 3, 1, 1, 1, False)
        torch.cuda.synchronize()
ops.tune(5, 0)
