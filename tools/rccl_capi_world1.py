"""The data-parallel step on RCCL's C API (csrc/comm.hip, auto_avsr_amd/comm.py) with ONE rank on one MI355X, full-size model by
default: (a) eager steps with `GradBuckets(comm=...)`, cross-rank BatchNorm through the same communicator and the W / sum(B)
all-gather -- losses against the same steps without any communicator; (b) the WHOLE step -- gradient buckets on their side
stream, 64 BatchNorm collectives, batch-size all-gather, fused optimizer -- captured into a hipGraph and replayed.  No torch
process group exists in this process: nothing polls events while the capture runs.  Prints one JSON line.
    python tools/rccl_capi_world1.py [--small] [--bucket-mb 64]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch

from synth import synth_batch, synth_state_dict

from auto_avsr_amd import functional as AF
from auto_avsr_amd.comm import StreamComm
from auto_avsr_amd.ddp import GradBuckets
from auto_avsr_amd.e2e import E2E
from auto_avsr_amd.optim import FusedAdamW

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
small = "--small" in sys.argv
bucket_mb = float(sys.argv[sys.argv.index("--bucket-mb") + 1]) if "--bucket-mb" in sys.argv else 64.0
V = 41 if small else 5049


def build():
    torch.manual_seed(0)
    m = (E2E(V, "video", adim=128, aheads=2, eunits=256, elayers=2, dunits=256, dlayers=2, cnn_module_kernel=7) if small
         else E2E(V, "video"))
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.load_state_dict(synth_state_dict(m.state_dict(), 31))
    return m.to(dev).train()


T = 8 if small else 24
x, lens, y = (t.to(dev) for t in synth_batch("video", 3, T, 3, V, seed=12, lengths=[8, 6, 5] if small else [24, 20, 17]))
out = {"model": "small" if small else "full 250M", "bucket_mb": bucket_mb}


def make(comm):
    AF.invalidate_weight_cache()
    AF.set_mode("bf16")
    AF.set_bn_sync(None if comm is None or os.environ.get("CAPI_NO_BNSYNC") == "1" else True,
                   comm=None if os.environ.get("CAPI_NO_BNSYNC") == "1" else comm)
    m = build()
    gb = None
    if comm is not None and os.environ.get("CAPI_NO_BUCKETS") != "1":
        gb = GradBuckets(m.parameters(), bucket_mb=bucket_mb, comm=None if os.environ.get("CAPI_NO_ALLREDUCE") == "1" else comm_grads)
    opt = FusedAdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0, warmup_steps=2,
                     total_steps=10, cast_weights=True)

    def step():
        if gb is not None and os.environ.get("CAPI_NO_BEGIN") != "1":
            gb.begin_step()
        AF.new_step()
        AF.refresh_weight_cache()
        loss = m.forward_tensors(x, lens, y)[0]
        if comm is not None:
            bs = torch.full((1,), float(x.shape[0]), device=dev)
            allb = torch.empty(comm.world, device=dev)
            comm.all_gather(allb, bs)
            loss = loss * (comm.world / allb.sum())
        else:
            loss = loss * (1.0 / x.shape[0])
        loss.backward()
        if gb is not None:
            gb.finish()
        if os.environ.get("CAPI_NO_OPT") != "1":
            opt.step()
        return loss

    return m, gb, opt, step


def eager(step, m, n=2):
    losses = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n):
            losses.append(float(step().detach()))
            for p in m.parameters():
                p.grad = None
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return losses


m0, _, _, step0 = make(None)
out["losses_plain"] = eager(step0, m0)
del m0, step0
comm, comm_grads = StreamComm.single(), StreamComm.single()  # BatchNorm / batch-size collectives; gradient buckets (side stream)
out["comm_world"] = comm.world
m1, gb, opt1, step1 = make(comm)
out["losses_capi"] = eager(step1, m1)
out["buckets"] = len(gb.flat) if gb is not None else 0
out["grad_wire"] = gb.wire if gb is not None else None  # AVSR_GRAD_WIRE=bf16: the buckets travel as bf16 (ncclBfloat16 sums)
assert all(abs(a - b) <= 2e-2 * abs(a) for a, b in zip(out["losses_plain"], out["losses_capi"])), out
print(json.dumps(dict(out, stage="before-capture")), flush=True)
AF.refresh_weight_cache()
torch.cuda.synchronize()
t0 = time.time()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gl = step1()
out["capture_seconds"] = round(time.time() - t0, 2)
before = [p.detach().clone() for p in m1.parameters()]
if os.environ.get("CAPI_DIAG") == "1":
    for _ in range(int(os.environ.get("CAPI_DIAG_REPLAYS", "1"))):
        g.replay()
    torch.cuda.synchronize()
    names = [n for n, _ in m1.named_parameters()]
    bad = [(names[i], int(torch.isnan(p.grad).sum()), p.grad.numel()) for i, p in enumerate(m1.parameters()) if p.grad is not None and torch.isnan(p.grad).any()]
    nog = [names[i] for i, p in enumerate(m1.parameters()) if p.grad is None]
    print(f"[diag] loss after the diag replays {float(gl)}; params with NaN grads: {len(bad)} of {len(names)}; first: {bad[:40]}; params without grad: {len(nog)}", flush=True)
    ok = [names[i] for i, p in enumerate(m1.parameters()) if p.grad is not None and not torch.isnan(p.grad).any()]
    print(f"[diag] params with finite grads ({len(ok)}): {ok[-40:]}", flush=True)
    inf = [(names[i], int(torch.isinf(p.grad).sum())) for i, p in enumerate(m1.parameters()) if p.grad is not None and torch.isinf(p.grad).any()]
    big = sorted(((float(p.grad.abs().max()), names[i]) for i, p in enumerate(m1.parameters()) if p.grad is not None and not torch.isnan(p.grad).any()), reverse=True)[:5]
    print(f"[diag] inf grads: {inf[:10]}; largest finite grads: {big}", flush=True)
    if gb is not None:
        print(f"[diag] NaN per bucket: {[int(torch.isnan(f).sum()) for f in gb.flat]}", flush=True)
        nv = sum(1 for i, p in enumerate(gb.params) if p.grad is not None and p.grad.data_ptr() == gb.views[i].data_ptr())
        print(f"[diag] grads that are bucket views: {nv} of {len(gb.params)}; captured tables {len(gb._captured)}", flush=True)
    wbad = [names[i] for i, p in enumerate(m1.parameters()) if torch.isnan(p).any()]
    print(f"[diag] params with NaN values after 1 replay: {len(wbad)} first {wbad[:8]}; opt state {opt1.state.tolist()[:3]}", flush=True)
t0 = time.time()
for r in range(5):
    g.replay()
    if os.environ.get("CAPI_DIAG") == "1":
        torch.cuda.synchronize()
        print(f"[diag] replay {r}: loss {float(gl):.5f} step/lr/gnorm {opt1.state.tolist()[:3]} NaN params {sum(int(torch.isnan(p).any()) for p in m1.parameters())}", flush=True)
torch.cuda.synchronize()
out["replay_ms"] = round((time.time() - t0) / 5 * 1e3, 2)
moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(m1.parameters(), before))
lg = float(gl.detach())
out.update(graph_loss=lg, graph_replay_moved_params=moved > 0, graph_capture="ok", opt_state=opt1.state.tolist()[:3],
           flags={k: v for k, v in os.environ.items() if k.startswith("CAPI_")})
print(json.dumps(dict(out, stage="after-replays")), flush=True)
assert lg == lg and abs(lg) < 1e30 and (moved > 0 or os.environ.get("CAPI_NO_OPT") == "1")
# after the replays the weights moved five optimizer steps: an eager step from here must still be finite (state intact)
for p in m1.parameters():
    p.grad = None
out["loss_eager_after_replays"] = eager(step1, m1, 1)[0]
assert out["loss_eager_after_replays"] == out["loss_eager_after_replays"]
AF.set_bn_sync(None)
comm.close()
comm_grads.close()
print(json.dumps(out), flush=True)
