"""Eager (no hipGraph) training step: host issue time vs device time.  The N>1 bench path is eager (DDP hooks), so the
host must stay ahead of the GPU.  GPU box:  python tools/eager_overhead.py [--prof]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from auto_avsr_amd import functional as AF
from auto_avsr_amd.e2e import E2E
from auto_avsr_amd.optim import FusedAdamW
from auto_avsr_amd.synthetic import bucket_batches, make_batch, rank_batches, utterance_lengths

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = E2E(5049, "video").to(dev).train()
AF.set_precise(False)
AF.manual_seed(1234)
seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
AF.set_seed_tensor(seed_dev)
opt = FusedAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0,
                 warmup_steps=5000, total_steps=75000)
lengths = utterance_lengths()
batches = rank_batches(bucket_batches(lengths, 1600, 400), 0, 1, seed=0)
x, lens, y, _ = make_batch(lengths, batches[len(batches) // 2], "video", 5049, seed=0, device=dev)
print("batch", tuple(x.shape), flush=True)


def step():
    AF.new_step()
    seed_dev.add_(1)
    AF.refresh_weight_cache()
    loss = model.forward_tensors(x, lens, y)[0]
    loss.backward()
    opt.step()
    opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
K = 8
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / K:.2f} ms/step   wall {1e3 * (t2 - t0) / K:.2f} ms/step", flush=True)
# host-only cost: same loop with the GPU kept idle-free is not possible; instead time fwd / bwd / opt issue separately
torch.cuda.synchronize()
tf = tb = to = 0.0
for _ in range(K):
    torch.cuda.synchronize()
    a = time.perf_counter()
    AF.new_step(); seed_dev.add_(1); AF.refresh_weight_cache()
    loss = model.forward_tensors(x, lens, y)[0]
    b = time.perf_counter()
    loss.backward()
    c = time.perf_counter()
    opt.step(); opt.zero_grad()
    d = time.perf_counter()
    tf += b - a; tb += c - b; to += d - c
print(f"issue (GPU drained before each step): fwd {1e3 * tf / K:.2f}  bwd {1e3 * tb / K:.2f}  opt {1e3 * to / K:.2f} ms", flush=True)
if "--prof" in sys.argv:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(4):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(35)
