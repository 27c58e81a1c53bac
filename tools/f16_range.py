"""Range check of the mixed mode's f16 activations on the MI355X: largest |x| of every f16 tensor the forward pass produces
(LayerNorm outputs, GEMM / convolution outputs, attention outputs, BatchNorm + SiLU outputs ...) over N training steps of the
bench workload (weights moving under AdamW 1e-3), against the f16 maximum 65504 (values beyond it are clamped on conversion,
prims.h f2h).  Prints one JSON line.
    python tools/f16_range.py [--steps 60]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from auto_avsr_amd import functional as AF
from auto_avsr_amd import ops
from auto_avsr_amd.e2e import E2E
from auto_avsr_amd.optim import FusedAdamW
from auto_avsr_amd.synthetic import bucket_batches, make_batch, rank_batches, utterance_lengths

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = E2E(5049, "video").to(dev).train()
AF.set_mode("mixed")
AF.manual_seed(1)
seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
AF.set_seed_tensor(seed_dev)
opt = FusedAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0, warmup_steps=20,
                 total_steps=1000, cast_weights=True)
lengths = utterance_lengths()
batches = rank_batches(bucket_batches(lengths, 1600, 400), 0, 1, seed=0)
seen = []
make = ops.TWIN


def spy(y):  # every f16 result that asks for a twin passes here
    if y.dtype == torch.float16:
        seen.append(y)
    return make(y)


ops.TWIN = spy
worst, losses, count = 0.0, [], 0
for step in range(args.steps):
    x, lens, y, _ = make_batch(lengths, batches[step % len(batches)], "video", 5049, seed=step, device=dev)
    AF.new_step()
    seed_dev.add_(1)
    AF.refresh_weight_cache()
    seen.clear()
    loss = model.forward_tensors(x, lens, y)[0]
    m = max(float(t.abs().max()) for t in seen)
    count = len(seen)
    worst = max(worst, m)
    seen.clear()
    loss.backward()
    opt.step()
    opt.zero_grad()
    losses.append(float(loss.detach()))
print(json.dumps({"steps": args.steps, "f16_tensors_per_step": count, "max_abs_f16_activation": worst, "f16_max": 65504.0,
                  "headroom": round(65504.0 / worst, 1), "loss_first": round(losses[0], 3), "loss_last": round(losses[-1], 3),
                  "all_finite": all(v == v and abs(v) < 1e30 for v in losses)}))
