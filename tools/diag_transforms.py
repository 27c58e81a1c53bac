"""Diagnostic: where does the device video transform differ from the CPU oracle?  GPU box: python tools/diag_transforms.py"""
import os, random, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import torch
import transforms_oracle as TO
from auto_avsr_amd import transforms as TR
dev = torch.device("cuda:0")
for subset in ("val", "train"):
    clip = torch.randint(0, 256, (61, 96, 96, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    torch.manual_seed(7); random.seed(7)
    want, crop, ivs = TO.video_transform(clip.permute(0, 3, 1, 2), subset)
    torch.manual_seed(7); random.seed(7)
    got = TR.VideoTransform(subset)(clip.to(dev).permute(0, 3, 1, 2)).cpu()
    d = (got - want).abs()
    bad = d > 0
    print(subset, "crop", crop, "ivs", ivs, "mismatching", int(bad.sum()), "of", bad.numel(), "max abs", float(d.max()),
          "frames with mismatch", bad.flatten(1).any(1).nonzero().flatten().tolist()[:20])
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        print("  first:", idx, float(got[tuple(idx)]), float(want[tuple(idx)]),
              "ulp diff", int(got[tuple(idx)].view(torch.int32)) - int(want[tuple(idx)].view(torch.int32)))
        ulps = (got.view(torch.int32) - want.view(torch.int32))[bad]
        print("  ulp histogram:", {int(k): int((ulps == k).sum()) for k in ulps.unique()[:10]})
