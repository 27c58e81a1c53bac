"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE need separate passes, MI355X_MICROARCH.md "rocprofv3
PMC slots"): a calibration stream of KNOWN byte counts, then ONE eager full training step of the bench workload.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o r -- python tools/pmc_step.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o r -- python tools/pmc_step.py
    python tools/pmc_report.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r2_hbm_traffic.txt

The calibration launches (scale_dropout_kernel<float, float> over 1 GiB: reads 1 GiB, writes 1 GiB, far beyond the 256 MiB
Infinity Cache) give the counter -> byte factors for this tool chain (the guide: FETCH_SIZE under-reports wide coalesced
reads 2x on gfx950, WRITE_SIZE is uncalibrated); pmc_report.py applies them.  AVSR_PMC_SHAPE=a|b picks the survey's
fixed batch A (4 x 400) or B (16 x 100); default: the middle bucketed batch of the bench workload."""
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

dev = torch.device("cuda:0")
ops.apply_env_tuning()
torch.manual_seed(0)
model = E2E(5049, "video").to(dev).train()
MODE = os.environ.get("AVSR_PMC_MODE", "mixed")  # the numerical mode bench.py times by default
AF.set_mode(MODE)
AF.manual_seed(1234)
seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
AF.set_seed_tensor(seed_dev)
opt = FusedAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0,
                 warmup_steps=5000, total_steps=75000, cast_weights=True)
lengths = utterance_lengths()
shape = os.environ.get("AVSR_PMC_SHAPE", "")
if shape == "a":
    lengths, idxs = [400, 380, 360, 340], [0, 1, 2, 3]
elif shape == "b":
    lengths, idxs = [100] * 16, list(range(16))
elif shape == "bench":
    # the batch bench.py's roofline legs re-issue their launches on: data[warmup] of the default run = shape 3 of the 8 it cycles
    batches = rank_batches(bucket_batches(lengths, 1600, 400), 0, 1, seed=0)
    by_len = sorted(batches, key=lambda b: max(int(lengths[i]) for i in b))
    idxs = by_len[(2 * 3 + 1) * len(by_len) // (2 * 8)]
else:
    batches = rank_batches(bucket_batches(lengths, 1600, 400), 0, 1, seed=0)
    idxs = batches[len(batches) // 2]
x, lens, y, frames = make_batch(lengths, idxs, "video", 5049, seed=0, device=dev)


def step():
    AF.new_step()
    seed_dev.add_(1)
    AF.refresh_weight_cache()
    loss = model.forward_tensors(x, lens, y)[0]
    loss.backward()
    opt.step()
    for p in model.parameters():
        p.grad = None
    return loss


for _ in range(2):  # warm-up: weight caches, allocator
    step()
torch.cuda.synchronize()
# ---- algorithmic work per C-ABI entry point of the same step (a step of its own, BEFORE the counters' region)
ops.TRACE = []
step()
torch.cuda.synchronize()
trace, ops.TRACE = ops.TRACE, None
per_entry = {}
for name, fl, nb in trace:
    e = per_entry.setdefault(name, [0, 0.0, 0.0])
    e[0] += 1
    e[1] += fl
    e[2] += nb
# ---- calibration: known bytes (marker kernels: the only scale_dropout_kernel<float, float> launches over 2^28 elements)
n = 1 << 28
src = torch.randn(n, device=dev)
for _ in range(3):
    dst = ops.scale_dropout(src, torch.float32, alpha=2.0)
torch.cuda.synchronize()
del src, dst
# ---- the measured step, bracketed by two tiny marker launches (sum_scale on 3 elements)
mark = torch.ones(3, device=dev)
ops.sum_scale(mark, 1.0)
loss = step()
ops.sum_scale(mark, 1.0)
torch.cuda.synchronize()
info = {"B": int(x.shape[0]), "T": int(x.shape[1]), "L": int(y.shape[2]), "real_frames": int(frames),
        "loss": float(loss.detach()),
        "shape": f"video, B={int(x.shape[0])} T={int(x.shape[1])} L={int(y.shape[2])}, {int(frames)} real frames, {MODE} mode, "
                 "full training step (fwd + bwd + clip + AdamW), eager launches",
        "entries": {k: {"calls": v[0], "flops": v[1], "bytes": v[2]} for k, v in per_entry.items()}}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(info, open(os.environ.get("AVSR_PMC_INFO", "gpurun_out/pmc_step_info.json"), "w"), indent=1)
print(json.dumps({k: info[k] for k in ("B", "T", "L", "real_frames", "loss")}))
