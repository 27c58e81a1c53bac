"""Few eager launches of the 3x3 weight-gradient kernel per trunk stage, for counter passes:
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d <dir> -- python tools/pmc_wgrad.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")
N = 1600
for variant in (1, 2):
    ops.tune(16, variant)
    for (H, C) in ((22, 64), (11, 128), (6, 256), (3, 512)):
        x = torch.randn(N, H, H, C, device=dev).bfloat16()
        dy = torch.randn(N, H, H, C, device=dev).bfloat16()
        for _ in range(3):
            ops.conv2d_wgrad(dy, x, N, H, H, C, C, 3, 3, 1, 1, 1, False, torch_layout=True)
        torch.cuda.synchronize()
