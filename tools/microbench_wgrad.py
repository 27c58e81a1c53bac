"""conv3x3_wgrad kernel alone at the four trunk stages (1600 frames), with its ablation knob (avsr_tune 5: 4 no staging after the first tile, 16 no tiles at all; knob 3 = 2: no output stage).  GPU box: python tools/microbench_wgrad.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from auto_avsr_amd import ops

dev = torch.device("cuda:0")


def t(fn, reps=20):
    """us per call, the calls replayed from a hipGraph (the Python side of one call costs more than the kernel)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


N = 1600
for variant, name in ((1, "thin: 4 waves of 32 co x 32 ci, two blocks per CU, 2-stage ring"),
                      (2, "fat: 4 waves of 64 co x 32 ci = 2 k groups x 2 ci halves, one block per CU, 3-stage ring"),
                      (3, "thin8: 8 thin waves = 2 k groups, one block per CU, 3-stage ring")):
  ops.tune(16, variant)
  print(name)
  for (H, C) in ((22, 64), (11, 128), (6, 256), (3, 512)):
      x = torch.randn(N, H, H, C, device=dev).bfloat16()
      dy = torch.randn(N, H, H, C, device=dev).bfloat16()
      fn = lambda: ops.conv2d_wgrad(dy, x, N, H, H, C, C, 3, 3, 1, 1, 1, False, torch_layout=True)
      flop = 2.0 * N * H * H * C * C * 9
      line = f"{H:2d}x{H:<2d} C={C:3d} ({flop / 1e9:5.1f} GFLOP):"
      for abl, name in ((0, "full"), (4, "nostage")):
          ops.tune(5, abl)
          us = t(fn)
          line += f"  {name} {us:6.1f}" + (f" ({flop / us / 1e6:4.0f} TF)" if abl == 0 else "")
      ops.tune(3, 2)  # no output stage: the blocks end after their last tile (the ordered reduce still runs, over stale partials)
      ops.tune(5, 4)
      line += f"  | no stores: nostage {t(fn):6.1f}"
      ops.tune(5, 16)
      line += f"  no tiles at all {t(fn):6.1f}"
      ops.tune(5, 0)
      line += f"  full {t(fn):6.1f}"
      ops.tune(3, 0)
      print(line, flush=True)
