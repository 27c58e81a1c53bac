"""The N > 1 code path on ONE MI355X: a single-rank RCCL process group (backend "nccl" = RCCL) so that everything bench.py /
train_native.py do for N > 1 -- process-group set-up, DDP with bucket views, the cross-rank BatchNorm collectives
(all_gather_into_tensor / all_reduce issued from inside the autograd functions), the W / sum(B) all-gather, FusedAdamW on
bucket-view gradients -- actually executes on RCCL before it meets an 8-GPU node (gpurun boxes have one GPU; RCCL refuses
two ranks on one device).  Checks the result against the same steps without any process group, then tries to capture the
distributed step into a hipGraph (collectives inside the capture) and reports whether this stack supports it.
Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import torch.distributed as dist

from synth import synth_batch, synth_state_dict

from auto_avsr_amd import functional as AF
from auto_avsr_amd.e2e import E2E
from auto_avsr_amd.optim import FusedAdamW

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
odim = 41
full = "--full" in sys.argv  # the 250 M-parameter model instead of the small instance


def build():
    torch.manual_seed(0)
    if full:
        m = E2E(5049, "video")
    else:
        m = E2E(odim, "video", adim=128, aheads=2, eunits=256, elayers=2, dunits=256, dlayers=2, cnn_module_kernel=7)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.load_state_dict(synth_state_dict(m.state_dict(), 31))
    return m.to(dev).train()


class Hot(torch.nn.Module):
    def __init__(self, mm):
        super().__init__()
        self.m = mm

    def forward(self, x, lens, y):
        return self.m.forward_tensors(x, lens, y)[0]


V = 5049 if full else odim
x, lens, y = (t.to(dev) for t in synth_batch("video", 3, 24 if full else 8, 3, V, seed=12, lengths=[24, 20, 17] if full else [8, 6, 5]))


def run(distributed, steps=2, precise=False):
    AF.invalidate_weight_cache()
    AF.set_precise(precise)
    AF.set_bn_sync(dist.group.WORLD if distributed else None)
    m = build()
    hot = Hot(m)
    if distributed:
        hot = torch.nn.parallel.DistributedDataParallel(hot, device_ids=[0], find_unused_parameters=False,
                                                        broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=64)
    opt = FusedAdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0, warmup_steps=2,
                     total_steps=10, cast_weights=True)
    losses, grads = [], None
    for _ in range(steps):
        AF.new_step()
        AF.refresh_weight_cache()
        loss = hot(x, lens, y)
        if distributed:
            bs = torch.tensor([float(x.shape[0])], device=dev)
            allb = torch.empty(1, device=dev)
            dist.all_gather_into_tensor(allb, bs)
            loss = loss * (1 / allb.sum()) * x.shape[0]  # world / sum(B) with world = 1, kept in the reference's form
        loss.backward()
        if grads is None:
            grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}
        opt.step()
        losses.append(float(loss.detach()))
        for p in m.parameters():
            p.grad = None
    AF.set_bn_sync(None)
    AF.set_precise(False)
    return m, hot, opt, losses, grads


os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "model": "full 250M" if full else "small"}
ONLY_BUCKETS = os.environ.get('WORLD1_ONLY_BUCKETS') == '1'
l2 = None
if not ONLY_BUCKETS:
    # (a) numerics: precise mode (split-bf16 contractions), first-step loss and gradients with and without the process group.
    # The distributed path takes other kernels for the BatchNorm statistics (partial sums -> all-gather -> Chan merge instead of
    # the single-rank finalize) and DDP averages through its buckets; with one rank both must reproduce the plain run.
    print('[world1] (a) plain precise', file=sys.stderr, flush=True)
    _, _, _, l0, g0 = run(False, steps=1, precise=True)
    torch.cuda.synchronize()
    print('[world1] (a2) rccl precise', file=sys.stderr, flush=True)
    _, _, _, l1, g1 = run(True, steps=1, precise=True)
    gmax = max(float(v.norm()) for v in g0.values())
    # Per-tensor agreement of the first-step gradients.  Tolerances: the split-bf16 contractions are deterministic but not
    # smooth (a 1e-7 relative change of a BatchNorm's invstd -- e.g. eps * (1 + 1e-6), no process group involved -- moves
    # individual FFN weight gradients of this synthetic-weight model by up to 1 %, tools/rccl_ab.py), and the distributed path
    # computes the BatchNorm statistics with other kernels (partial sums -> all-gather -> Chan merge).  Gradients that are
    # analytically zero (a bias in front of a train-mode BatchNorm, linear_k.bias) are rounding noise on both sides: skipped.
    cos, ratio = {}, {}
    for k in g0:
        n0 = float(g0[k].norm())
        if n0 < 1e-3 * gmax:
            continue
        cos[k] = float(torch.dot(g0[k].flatten(), g1[k].flatten())) / (n0 * float(g1[k].norm()) + 1e-30)
        ratio[k] = float(g1[k].norm()) / n0
    wk = min(cos, key=cos.get)
    out.update(loss_plain=l0[0], loss_rccl=l1[0], grad_tensors_compared=len(cos), grad_cos_min=cos[wk], grad_cos_min_tensor=wk,
               grad_norm_ratio_range=[min(ratio.values()), max(ratio.values())])
    assert abs(l0[0] - l1[0]) <= 1e-5 * abs(l0[0]), (l0, l1)
    assert cos[wk] > 0.999 and 0.98 < min(ratio.values()) and max(ratio.values()) < 1.02, (wk, cos[wk], out["grad_norm_ratio_range"])
    # (b) the bench configuration: bf16 mode, two full steps incl. FusedAdamW(cast_weights) on bucket-view gradients
    torch.cuda.synchronize()
    print('[world1] (b) bf16 ddp', file=sys.stderr, flush=True)
    m1, hot1, opt1, l2, _ = run(True, steps=2)
    torch.cuda.synchronize()
    out.update(losses_bf16_rccl=l2)
    assert all(v == v and abs(v) < 1e30 for v in l2)


# ---- the same step on this build's own gradient exchange (auto_avsr_amd/ddp.py GradBuckets) instead of torch DDP:
# eager numerics against the torch-DDP run above, then hipGraph capture + replay WITH the RCCL collectives inside
try:
    from auto_avsr_amd.ddp import GradBuckets

    torch.cuda.synchronize()
    print('[world1] (c) grad buckets eager', file=sys.stderr, flush=True)
    AF.set_bn_sync(dist.group.WORLD if os.environ.get('WORLD1_NO_BNSYNC') != '1' else None)
    m2 = build()
    gb = GradBuckets(m2.parameters(), group=dist.group.WORLD if os.environ.get('WORLD1_NO_ALLREDUCE') != '1' else None, bucket_mb=float(os.environ.get('WORLD1_BUCKET_MB', '64')))
    opt2 = FusedAdamW(m2.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.03, max_grad_norm=10.0, warmup_steps=2,
                      total_steps=10, cast_weights=os.environ.get('WORLD1_NO_CAST') != '1')
    if os.environ.get('WORLD1_NO_BUCKETS') == '1':
        gb.remove()
    AF.invalidate_weight_cache()

    def step2():
        AF.new_step()
        AF.refresh_weight_cache()
        loss = m2.forward_tensors(x, lens, y)[0]
        bs = torch.full((1,), float(x.shape[0]), device=dev)
        allb = torch.empty(1, device=dev)
        dist.all_gather_into_tensor(allb, bs)
        loss = loss * (1 / allb.sum()) * x.shape[0]
        loss.backward()
        if os.environ.get('WORLD1_NO_BUCKETS') != '1':
            gb.finish()
        if os.environ.get('WORLD1_NO_OPT') != '1':
            opt2.step()
        return loss

    l3 = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            l3.append(float(step2().detach()))
            for p in m2.parameters():
                p.grad = None
    torch.cuda.current_stream().wait_stream(side)
    AF.refresh_weight_cache()
    torch.cuda.synchronize()
    out.update(losses_bf16_grad_buckets=l3, buckets=len(gb.flat))
    assert l2 is None or abs(l3[0] - l2[0]) <= 2e-2 * abs(l2[0]), (l3, l2)  # same first step as the torch-DDP run (bf16 mode, same seed)
    print('[world1] (d) grad buckets capture', file=sys.stderr, flush=True)
    print(json.dumps(dict(out, stage='before-capture')), flush=True)  # (a capture can abort the process: torch's watchdog thread)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
        gl2 = step2()
    before = [p.detach().clone() for p in m2.parameters()]
    if os.environ.get('WORLD1_DIAG') == '1':
        torch.cuda.synchronize()
        nview = sum(1 for i, p in enumerate(gb.params) if p.grad is not None and p.grad.data_ptr() == gb.views[i].data_ptr())
        print(f"[diag] params {len(gb.params)} grads-that-are-views {nview} buckets {len(gb.flat)} opt tables {len(opt2._tables)} "
              f"(captured flags {[e[5] for e in opt2._tables.values()]}) bucket tables {len(gb._tables)} "
              f"state before replay {opt2.state.tolist()}", file=sys.stderr, flush=True)
        g2.replay()
        torch.cuda.synchronize()
        print(f"[diag] state after 1 replay {opt2.state.tolist()} flat abs sums {[float(f.abs().sum()) for f in gb.flat][:6]}",
              file=sys.stderr, flush=True)
    for _ in range(3):
        g2.replay()
    torch.cuda.synchronize()
    moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(m2.parameters(), before))
    lg = float(gl2.detach())
    out.update(graph_capture_with_grad_buckets="ok", graph_loss_grad_buckets=lg, graph_replay_moved_params_grad_buckets=moved > 0)
    assert lg == lg and moved > 0
except Exception as e:
    import traceback

    traceback.print_exc()
    out.update(graph_capture_with_grad_buckets=f"failed: {type(e).__name__}: {str(e)[:300]}")
    try:
        torch.cuda.synchronize()
    except Exception:
        pass
# ---- can the distributed step be captured into a hipGraph (RCCL collectives as graph nodes)?
try:
    if ONLY_BUCKETS:
        raise RuntimeError('skipped (WORLD1_ONLY_BUCKETS)')
    AF.set_bn_sync(dist.group.WORLD)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())

    def step():
        AF.new_step()
        AF.refresh_weight_cache()
        loss = hot1(x, lens, y)
        loss.backward()
        opt1.step()
        return loss

    with torch.cuda.stream(side):
        for _ in range(3):  # DDP wants its first iterations outside capture (bucket rebuild)
            step()
            for p in m1.parameters():
                p.grad = None
    torch.cuda.current_stream().wait_stream(side)
    AF.refresh_weight_cache()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    # thread-local capture mode: the process group's watchdog thread queries events while we capture
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        gl = step()
    before = [p.detach().clone() for p in m1.parameters()]
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(m1.parameters(), before))
    out.update(graph_capture_with_rccl="ok", graph_loss=float(gl.detach()), graph_replay_moved_params=moved > 0)
except Exception as e:  # reported, not fatal: the eager N > 1 path above is what bench.py uses
    out.update(graph_capture_with_rccl=f"failed: {type(e).__name__}: {str(e)[:300]}")
AF.set_bn_sync(None)
print(json.dumps(out), flush=True)
try:
    dist.destroy_process_group()
except Exception:
    pass
