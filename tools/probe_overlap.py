"""GPU probe (round 6): how much of the optimizer step (HBM-bound, 1.7 ms) hides under the NEXT step's front-end forward pass
(MFMA / LDS-bound, ~5 ms) when the two run as separate hipGraphs on two streams?  Batch A of the video bench, mixed mode.
Prints: front-end forward graph alone, optimizer graph alone, both launched together (wall clock of the pair)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_avsr_amd import functional as AF
from auto_avsr_amd.e2e import E2E
from auto_avsr_amd.optim import FusedAdamW
from auto_avsr_amd.synthetic import make_batch

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = E2E(5049, "video").to(dev).train()
AF.set_mode("mixed")
AF.manual_seed(1)
seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
AF.set_seed_tensor(seed_dev)
opt = FusedAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0, warmup_steps=5000,
                 total_steps=75000, cast_weights=True)
lengths = [400, 380, 360, 340]
x, lens, y, _ = make_batch(lengths, list(range(4)), "video", 5049, seed=0, device=dev)
y = y[:, :, :64].contiguous()
params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    AF.new_step()
    seed_dev.add_(1)
    AF.refresh_weight_cache()
    loss = model.forward_tensors(x, lens, y)[0]
    loss.backward()
    opt.step()


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    for _ in range(3):
        step()
    for p in params:
        p.grad = None
    AF.new_step()
    AF.refresh_weight_cache()
    loss = model.forward_tensors(x, lens, y)[0]
    loss.backward()  # gradients stay alive for the optimizer graph
torch.cuda.synchronize()

g_opt = torch.cuda.CUDAGraph()
with torch.cuda.stream(s2):
    with torch.cuda.graph(g_opt, stream=s2):
        opt.step()
g_front = torch.cuda.CUDAGraph()
with torch.cuda.stream(s1):
    AF.new_step()
    AF.refresh_weight_cache()
    with torch.cuda.graph(g_front, stream=s1):
        AF.new_step()
        AF.refresh_weight_cache()
        feats = model.frontend(x)
torch.cuda.synchronize()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(torch.cuda.current_stream())
    for _ in range(reps):
        fn()
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cur = torch.cuda.current_stream()


def front_only():
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        g_front.replay()
    cur.wait_stream(s1)


def opt_only():
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        g_opt.replay()
    cur.wait_stream(s2)


def both():
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        g_opt.replay()
    with torch.cuda.stream(s1):
        g_front.replay()
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def serial():
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        g_opt.replay()
        g_front.replay()
    cur.wait_stream(s1)


for name, fn in (("front-end forward alone", front_only), ("optimizer alone", opt_only), ("one after the other", serial),
                 ("together on two streams", both)):
    print(f"{name:28s} {timed(fn):7.3f} ms", flush=True)

# ---- the same with the optimizer's stream restricted to a subset of the CUs (hipExtStreamCreateWithCUMask): a deterministic split
# of the chip instead of the dispatcher's time slicing
import ctypes

hip = ctypes.CDLL("libamdhip64.so")
for label, bits in (("every 8th CU (32)", [i for i in range(256) if i % 8 == 0]), ("every 4th CU (64)", [i for i in range(256) if i % 4 == 0]),
                    ("CUs 0..31", list(range(32))), ("CUs 0..63", list(range(64)))):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    if rc != 0:
        print(f"hipExtStreamCreateWithCUMask rc={rc}")
        continue
    s2 = torch.cuda.ExternalStream(h.value)
    print(f"optimizer stream on {label}: alone {timed(opt_only):7.3f} ms   together {timed(both):7.3f} ms", flush=True)
    words_f = (ctypes.c_uint32 * 8)()
    for b in range(256):
        if b not in set(bits):
            words_f[b // 32] |= 1 << (b % 32)
    hf = ctypes.c_void_p()
    if hip.hipExtStreamCreateWithCUMask(ctypes.byref(hf), 8, words_f) == 0:
        s1_old, s1 = s1, torch.cuda.ExternalStream(hf.value)
        print(f"   ... and the front end on the complement: alone {timed(front_only):7.3f} ms   together {timed(both):7.3f} ms", flush=True)
        s1 = s1_old
